#!/bin/bash
# does the persistent Winograd kernel (2 workgroups per CU, 246 registers: nothing else fits on a SIMD while it runs) starve the main
# chain's small latency-bound launches?  headline with its grid cut from 512 to fewer workgroups (tools build: JM_WN_GRID).
#   gpurun -- 'bash tools/wino_grid_sweep.sh'
for g in 512 480 448 384 320 256; do
  for w in detect train; do
    extra=""; [ $w = train ] && extra="--workload train"
    JM_WN_GRID=$g python tools/ab_lib.py tools/bin/libjmodt_hip_tools.so bench.py --no-cpu-baseline --headline-only $extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('grid $g $w', d['value'], d['ms_per_step'], (d.get('image_branch_kernel') or {}).get('ms_per_step'))"
  done
done
