#!/bin/bash
# tools/sa_exp.sh — build variants of the tools library whose fused-SA inner block leaves one operand feed out
# (jm_mfma.h, JM_SA_EXP bits) -> tools/bin/libjmodt_hip_tools_exp<N>.so; time them with tools/sa_exp.py on the GPU box.
set -e
cd "$(dirname "$0")/.."
python -m jmodt_amd.csrc.build --tools > /dev/null
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden -munsafe-fp-atomics -DJM_BUILDING -DJM_TOOLS_BUILD -I jmodt_amd/csrc -I include -w"
for n in "$@"; do
  /opt/rocm/bin/hipcc $FLAGS -DJM_SA_EXP=$n -c jmodt_amd/csrc/sa_mlp.hip -o /tmp/sa_mlp_exp$n.o
  objs=$(ls jmodt_amd/csrc/build_tools/*.o | grep -v '/sa_mlp.o')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/bin/libjmodt_hip_tools_exp$n.so $objs /tmp/sa_mlp_exp$n.o
  echo built exp$n
done
