#!/bin/bash
# Run ON THE GPU BOX: rocprofv3 kernel-trace stats of the joint-mode training step.
#   gpurun --timeout 1500 -- 'bash tools/profile_joint.sh r05_joint_before'
set -u
TAG=${1:-r05_joint}
shift || true
REPO=$(pwd)
OUT=$REPO/gpurun_out/profiles
mkdir -p "$OUT"
export TMPDIR=/tmp
export MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD=0
cd /tmp
timeout 900 python "$REPO/bench.py" --workload train --joint --steps 3 --warmup 2 --no-cpu-baseline "$@" > /tmp/warm.log 2>&1
tail -1 /tmp/warm.log | cut -c1-400
rm -rf /tmp/prof_$TAG
timeout 900 rocprofv3 --stats --kernel-trace -d /tmp/prof_$TAG -o $TAG -- python "$REPO/bench.py" --workload train --joint --steps 6 --warmup 2 --no-cpu-baseline "$@" > /tmp/prof_$TAG.log 2>&1
db=$(find /tmp/prof_$TAG -name '*.db' | head -1)
if [ -z "$db" ]; then echo "no db"; tail -5 /tmp/prof_$TAG.log; exit 1; fi
python "$REPO/profiles/summarize.py" "$db" "$OUT/${TAG}_kernel_stats.txt" > /dev/null
tail -1 /tmp/prof_$TAG.log | cut -c1-600
mkdir -p "$REPO/gpurun_out/bench_out"; cp /tmp/bench_out/*.json "$REPO/gpurun_out/bench_out/" 2>/dev/null; cp "$REPO"/bench_out/*.json "$REPO/gpurun_out/bench_out/" 2>/dev/null
head -70 "$OUT/${TAG}_kernel_stats.txt"
