#!/bin/bash
# Run ON THE GPU BOX: the whole GPU tier, smoke(), the driver's line, the RCNN-step variants
OUT=gpurun_out/r06_final
mkdir -p $OUT
export MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD=0
timeout 1200 python -m pytest tests -m gpu -q --durations=8 > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -14 $OUT/pytest.log | cut -c1-200
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'])"; }
timeout 300 python bench.py 2>/dev/null | grep "^{" | tee $OUT/default.line.json | line default
timeout 300 python bench.py --workload train --rcnn --no-cpu-baseline --headline-only --steps 30 --warmup 5 2>/dev/null | grep "^{" | line rcnn
timeout 300 python bench.py --workload train --rcnn --no-cpu-baseline --headline-only --steps 30 --warmup 5 --prefetch-depth 2 2>/dev/null | grep "^{" | line rcnn_depth2
