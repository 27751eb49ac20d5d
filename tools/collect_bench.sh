#!/bin/bash
# the bench lines of the current tree, one JSON file per workload (copied into profiles/bench_rNN/ afterwards)
#   gpurun --timeout 2400 -- 'bash tools/collect_bench.sh'
OUT=gpurun_out/bench_lines
mkdir -p $OUT
python bench.py 2>$OUT/default.err | grep "^{" > $OUT/default.json
for w in train sa ops dense dense_detect; do
    python bench.py --workload $w --no-cpu-baseline 2>$OUT/$w.err | grep "^{" > $OUT/$w.json
done
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --workload train --no-cpu-baseline 2>$OUT/train_launch.err | grep "^{" > $OUT/train_launch.json
for f in $OUT/*.json; do python - "$f" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[1].split("/")[-1], d["value"], d["unit"], d["ms_per_step"], d.get("roofline", {}).get("kernel"), d.get("roofline", {}).get("frac"))
PY
done
