#!/bin/bash
# Run ON THE GPU BOX: the GPU tier, then the driver's bench line and the two training lines of the current tree.
#   gpurun --timeout 1500 -- 'bash tools/r06_check.sh'
OUT=gpurun_out/r06_check
mkdir -p $OUT
export MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD=0
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" ; tail -3 $OUT/pytest.log
timeout 300 python bench.py --full-out $OUT/default.json 2>$OUT/default.err | grep "^{" > $OUT/default.line.json; cut -c1-260 $OUT/default.line.json
timeout 300 python bench.py --workload train --rcnn --no-cpu-baseline --full-out $OUT/train_rcnn.json 2>$OUT/train_rcnn.err | grep "^{" > $OUT/train_rcnn.line.json; cut -c1-400 $OUT/train_rcnn.line.json; tail -3 $OUT/train_rcnn.err
timeout 300 python bench.py --workload train --joint --no-cpu-baseline --steps 6 --warmup 2 --full-out $OUT/train_joint.json 2>$OUT/train_joint.err | grep "^{" > $OUT/train_joint.line.json; cut -c1-400 $OUT/train_joint.line.json; tail -3 $OUT/train_joint.err
