OUT=gpurun_out/bench_lines
python bench.py 2>$OUT/default.err | grep "^{" > $OUT/default.json
python bench.py --workload train --no-cpu-baseline 2>$OUT/train.err | grep "^{" > $OUT/train.json
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --workload train --no-cpu-baseline 2>$OUT/train_launch.err | grep "^{" > $OUT/train_launch.json
for f in default train train_launch; do python -c "
import json,sys; d=json.load(open('$OUT/$f.json')); print('$f', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], (d.get('image_branch_kernel') or {}).get('frac'), (d.get('image_branch_kernel') or {}).get('direct_form'))"; done
