#!/bin/bash
# Run ON THE GPU BOX: host-side profile of the two training steps, round 6's rocprofv3 passes, the MIOpen over-read probe (channels-last)
OUT=gpurun_out/r06_evidence
mkdir -p $OUT
export MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD=0
for m in joint rcnn; do timeout 400 python tools/joint_host_profile.py $m 5 > $OUT/host_profile_$m.txt 2>&1; head -3 $OUT/host_profile_$m.txt | grep host; done
bash tools/collect_profiles_r06.sh r06 all 2>&1 | tail -40
for what in dgrad fwd; do for kc in "8 16" "32 64"; do
  n="oob_${what}_nhwc_${kc// /_}"
  timeout 120 python -X faulthandler tools/miopen_oob_probe.py $what nhwc $kc > $OUT/$n.log 2>&1; rc=$?
  echo "$n rc=$rc $(grep -a -E 'Memory access|^OK' $OUT/$n.log | cut -c1-200 | tr '\n' '|')"
done; done
timeout 60 python -c "import torch; x=torch.ones(8,device='cuda'); print('gpu alive', float(x.sum()))"
