#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 kernel-trace stats + PMC passes (one counter per pass, never combined
# with other trace domains) for the bench workloads.  The rocpd databases stay in /tmp (they exceed gpurun's 64 MiB
# copy-back limit); only the text summaries land in gpurun_out/profiles/, from where they are copied into profiles/.
#   gpurun --timeout 2400 -- 'bash tools/collect_profiles.sh r04'
set -u
ROUND=${1:-r04}
REPO=$(pwd)
OUT=$REPO/gpurun_out/profiles
mkdir -p "$OUT"
export TMPDIR=/tmp
# MIOpen's find mode (the engine scopes it to the image convolutions) also times its reference kernel, 0.15 s per call and
# shape, at start-up: keep those probes out of the traces
export MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD=0
cd /tmp
run() {   # tag, extra rocprof flags, summarize mode, bench args...
    local tag=$1 flags=$2 mode=$3; shift 3
    rm -rf /tmp/prof_$tag
    timeout 900 rocprofv3 $flags --kernel-trace -d /tmp/prof_$tag -o $tag -- python "$REPO/bench.py" "$@" > /tmp/prof_$tag.log 2>&1
    local db; db=$(find /tmp/prof_$tag -name '*.db' | head -1)
    if [ -z "$db" ]; then echo "no db for $tag"; tail -5 /tmp/prof_$tag.log; return; fi
    python "$REPO/profiles/summarize.py" $mode "$db" "$OUT/${ROUND}_$tag.txt" > /dev/null
    tail -1 /tmp/prof_$tag.log | cut -c1-300
}
# MIOpen tunes every new convolution shape on first use (seconds of naive_conv_* kernels on a fresh box): do that outside the profile
timeout 600 python "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > /tmp/warm.log 2>&1
run detect_kernel_stats "--stats" ""   --steps 40 --warmup 3 --no-cpu-baseline --headline-only
run sa_kernel_stats     "--stats" ""   --workload sa --steps 12 --warmup 3 --no-cpu-baseline
run ops_kernel_stats    "--stats" ""   --workload ops --steps 7 --warmup 2 --no-cpu-baseline
for c in FETCH_SIZE WRITE_SIZE MfmaUtil SQ_INSTS_VALU_MFMA_MOPS_F32; do
    run detect_pmc_$c "--pmc $c" "--pmc" --steps 2 --warmup 1 --no-cpu-baseline --headline-only
done
for c in FETCH_SIZE WRITE_SIZE; do
    run sa_pmc_$c  "--pmc $c" "--pmc" --workload sa --steps 3 --warmup 1 --no-cpu-baseline
    run ops_pmc_$c "--pmc $c" "--pmc" --workload ops --steps 3 --warmup 1 --no-cpu-baseline
done
ls -la "$OUT"
