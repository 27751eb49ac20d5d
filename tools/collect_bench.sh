#!/bin/bash
# the bench lines of the current tree, one JSON file per workload (copied into profiles/bench_rNN/ afterwards)
#   gpurun --timeout 2400 -- 'bash tools/collect_bench.sh'
OUT=gpurun_out/bench_lines
mkdir -p $OUT
# <name>.json = the FULL record (kernel table, variants, parity block), <name>.line.json = the compact stdout line
python bench.py --full-out $OUT/default.json 2>$OUT/default.err | grep "^{" > $OUT/default.line.json
python bench.py --workload sa --full-out $OUT/sa.json 2>$OUT/sa.err | grep "^{" > $OUT/sa.line.json
for w in train ops dense dense_detect; do
    python bench.py --workload $w --no-cpu-baseline --full-out $OUT/$w.json 2>$OUT/$w.err | grep "^{" > $OUT/$w.line.json
done
for c in kitti packed; do
    python bench.py --cloud $c --no-cpu-baseline --full-out $OUT/detect_$c.json 2>$OUT/detect_$c.err | grep "^{" > $OUT/detect_$c.line.json
done
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --workload train --no-cpu-baseline --full-out $OUT/train_launch.json 2>$OUT/train_launch.err | grep "^{" > $OUT/train_launch.line.json
python bench.py --workload train --joint --launch --no-cpu-baseline --steps 6 --warmup 2 --full-out $OUT/train_joint_launch.json 2>$OUT/train_joint_launch.err | grep "^{" > $OUT/train_joint_launch.line.json
# the reference's default training mode (RPN fixed): plain, under the launcher (one-rank RCCL group), without the frozen half issued ahead
python bench.py --workload train --rcnn --no-cpu-baseline --full-out $OUT/train_rcnn.json 2>$OUT/train_rcnn.err | grep "^{" > $OUT/train_rcnn.line.json
python bench.py --workload train --rcnn --launch --no-cpu-baseline --full-out $OUT/train_rcnn_launch.json 2>$OUT/train_rcnn_launch.err | grep "^{" > $OUT/train_rcnn_launch.line.json
python bench.py --workload train --rcnn --no-ahead --no-cpu-baseline --headline-only --full-out $OUT/train_rcnn_no_ahead.json 2>$OUT/train_rcnn_no_ahead.err | grep "^{" > $OUT/train_rcnn_no_ahead.line.json
for f in $OUT/*.line.json; do python - "$f" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[1].split("/")[-1], d["value"], d["unit"], d["ms_per_step"], (d.get("roofline") or {}).get("kernel"), (d.get("roofline") or {}).get("frac"))
PY
done
