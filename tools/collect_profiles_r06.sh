#!/bin/bash
# Run ON THE GPU BOX (through gpurun): round 6's additions to tools/collect_profiles.sh —
#   * the two training steps (--joint, --rcnn): kernel stats + one PMC pass each for MfmaUtil / FETCH_SIZE / WRITE_SIZE, so that the
#     rows_* kernels' fractions can be recomputed from profiles/ (VERDICT r5 weak #6 / next #3);
#   * the composed detect step on the PACKED cloud: SQ wave / stall / LDS counters and the L2 hit counters for the kernels that lag
#     there (rcnn_lift, sa_mlp_pm listed, sa_mlp listed) — "why", measured instead of argued.
# PMC passes never share a run with another trace domain; SQ counters of one pass fit the 8 SQ slots.
#   gpurun --timeout 2400 -- 'bash tools/collect_profiles_r06.sh r06 [joint|ops|all]'
set -u
ROUND=${1:-r06}
WHAT=${2:-all}
REPO=$(pwd)
OUT=$REPO/gpurun_out/profiles
mkdir -p "$OUT"
export TMPDIR=/tmp
export MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD=0
cd /tmp
rocprofv3 -L > "$OUT/${ROUND}_counters_available.txt" 2>&1
have() { local ok=""; for c in "$@"; do if grep -q -w "$c" "$OUT/${ROUND}_counters_available.txt"; then ok="$ok $c"; else echo "counter $c: not offered by this rocprofv3" >&2; fi; done; echo $ok; }
run() {   # tag, extra rocprof flags, summarize mode, bench args...
    local tag=$1 flags=$2 mode=$3; shift 3
    rm -rf /tmp/prof_$tag
    timeout 900 rocprofv3 $flags --kernel-trace -d /tmp/prof_$tag -o $tag -- python "$REPO/bench.py" "$@" > /tmp/prof_$tag.log 2>&1
    local db; db=$(find /tmp/prof_$tag -name '*.db' | head -1)
    if [ -z "$db" ]; then echo "no db for $tag"; tail -5 /tmp/prof_$tag.log; return; fi
    python "$REPO/profiles/summarize.py" $mode "$db" "$OUT/${ROUND}_$tag.txt" > /dev/null
    echo "$tag: $(tail -1 /tmp/prof_$tag.log | cut -c1-200)"
}
if [ "$WHAT" = all ] || [ "$WHAT" = joint ]; then
    for mode in joint rcnn; do
        timeout 600 python "$REPO/bench.py" --workload train --$mode --steps 2 --warmup 2 --no-cpu-baseline --headline-only > /tmp/warm_$mode.log 2>&1
        run ${mode}_kernel_stats "--stats" "" --workload train --$mode --steps 8 --warmup 3 --no-cpu-baseline --headline-only
        db=$(find /tmp/prof_${mode}_kernel_stats -name '*.db' | head -1)
        [ -n "$db" ] && python "$REPO/tools/joint_timeline.py" "$db" "$OUT/${ROUND}_${mode}_timeline.csv" | cut -c1-160
        for c in MfmaUtil FETCH_SIZE WRITE_SIZE; do
            run ${mode}_pmc_$c "--pmc $c" "--pmc" --workload train --$mode --steps 2 --warmup 2 --no-cpu-baseline --headline-only
        done
    done
fi
if [ "$WHAT" = all ] || [ "$WHAT" = ops ]; then
    timeout 600 python "$REPO/bench.py" --cloud packed --steps 2 --warmup 1 --no-cpu-baseline --headline-only > /tmp/warm_packed.log 2>&1
    run packed_kernel_stats "--stats" "" --cloud packed --steps 20 --warmup 3 --no-cpu-baseline --headline-only
    A=$(have SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS)
    B=$(have SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM)
    C=$(have SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS)
    D=$(have TCC_HIT_sum TCC_MISS_sum)
    E=$(have TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum)
    [ -n "$A" ] && run packed_pmc_sq_waves   "--pmc $A" "--pmc" --cloud packed --steps 2 --warmup 1 --no-cpu-baseline --headline-only
    [ -n "$B" ] && run packed_pmc_sq_insts   "--pmc $B" "--pmc" --cloud packed --steps 2 --warmup 1 --no-cpu-baseline --headline-only
    [ -n "$C" ] && run packed_pmc_sq_lds     "--pmc $C" "--pmc" --cloud packed --steps 2 --warmup 1 --no-cpu-baseline --headline-only
    [ -n "$D" ] && run packed_pmc_l2_hit     "--pmc $D" "--pmc" --cloud packed --steps 2 --warmup 1 --no-cpu-baseline --headline-only
    [ -n "$E" ] && run packed_pmc_l1         "--pmc $E" "--pmc" --cloud packed --steps 2 --warmup 1 --no-cpu-baseline --headline-only
fi
ls -la "$OUT" | tail -30
