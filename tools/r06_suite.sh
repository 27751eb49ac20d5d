#!/bin/bash
# Run ON THE GPU BOX: the whole GPU tier with durations, then the MIOpen over-read probe (evidence for DESIGN.md section 6)
OUT=gpurun_out/r06_suite
mkdir -p $OUT
export MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD=0
timeout 1500 python -m pytest tests -m gpu -q --durations=25 > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -35 $OUT/pytest.log
for what in fwd wgrad dgrad; do
  timeout 120 python -X faulthandler tools/miopen_oob_probe.py $what nchw 8 16 > $OUT/oob_$what.log 2>&1; rc=$?
  echo "oob_$what rc=$rc $(grep -a -E 'Memory access|^OK|tail slot' $OUT/oob_$what.log | cut -c1-200 | tr '\n' '|')"
done
timeout 60 python -c "import torch; x=torch.ones(8,device='cuda'); print('gpu alive', float(x.sum()))"
