#!/bin/bash
# frames/s of the sa / train / detect workloads with the FPS pyramids of 0..3 upcoming batches in flight, at the default number of
# hardware queues and with GPU_MAX_HW_QUEUES=8.   gpurun -- 'bash tools/prefetch_depth_sweep.sh'
mkdir -p gpurun_out/r04/pd
run() { name=$1; shift; python bench.py --no-cpu-baseline --headline-only "$@" > gpurun_out/r04/pd/$name.json 2> gpurun_out/r04/pd/$name.err; python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r04/pd/$name.json').read().strip().splitlines()[-1])
    print('$name', d['value'], d['ms_per_step'], d.get('host_enqueue_ms_per_step'), d.get('fps',{}).get('chain_ms_per_step'))
except Exception as e:
    print('$name', 'FAILED', e)
PY
}
for q in default 8; do
  if [ $q != default ]; then export GPU_MAX_HW_QUEUES=$q; fi
  run q${q}_sa_d0 --workload sa --no-prefetch
  for d in 1 2 3; do run q${q}_sa_d$d --workload sa --prefetch-depth $d; done
  for d in 1 2; do run q${q}_train_d$d --workload train --prefetch-depth $d; done
  for d in 1 2; do run q${q}_detect_d$d --prefetch-depth $d; done
done
