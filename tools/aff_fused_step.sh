#!/bin/bash
# headline with the link head on the two-launch chain / the one-kernel form (tools build: JM_AFF_FUSED; JM_AFF_GRID: -1 = one
# workgroup per tile (default), n = persistent on n workgroups).   gpurun -- 'bash tools/aff_fused_step.sh'
run() { JM_AFF_FUSED=$1 JM_AFF_GRID=$2 timeout 250 python tools/ab_lib.py tools/bin/libjmodt_hip_tools.so bench.py --no-cpu-baseline --headline-only 2>/dev/null < /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('fused $1 grid $2', d['value'], d['ms_per_step'], r['frac'], r.get('avg_launch_ms'), (r.get('isolated') or {}).get('frac'))"; }
run 0 -1; run 1 -1; run 0 -1; run 1 -1; run 1 256
