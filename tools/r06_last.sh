#!/bin/bash
# Run ON THE GPU BOX: the whole GPU tier on the final tree, smoke(), and the joint line of the OPERATOR route (train-mode BatchNorm:
# the form the reference's joint mode trains; ADVICE r5 #1 asks for both BatchNorm modes side by side)
OUT=gpurun_out/r06_last
mkdir -p $OUT gpurun_out/bench_lines
export MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD=0
timeout 1200 python -m pytest tests -m gpu -q --durations=5 > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -12 $OUT/pytest.log | cut -c1-200
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
JM_JOINT_ROUTE=operators timeout 1500 python bench.py --workload train --joint --no-cpu-baseline --headline-only --steps 6 --warmup 2 --full-out gpurun_out/bench_lines/train_joint_operators.json 2>$OUT/ops.err | grep "^{" > gpurun_out/bench_lines/train_joint_operators.line.json
python -c "import json; d=json.load(open('gpurun_out/bench_lines/train_joint_operators.line.json')); print('joint operators', d['value'], d['ms_per_step'], d['config'].get('batchnorm','')[:60])"
timeout 600 python bench.py --workload train --joint --no-cpu-baseline --full-out gpurun_out/bench_lines/train_joint.json 2>$OUT/joint.err | grep "^{" > gpurun_out/bench_lines/train_joint.line.json
python -c "import json; d=json.load(open('gpurun_out/bench_lines/train_joint.line.json')); print('joint rows', d['value'], d['ms_per_step'], d.get('clouds'))"
