#!/bin/bash
# timeline of gemm1x1_w4_kernel with the timing ablations of the profiling library (results WRONG by design):
# LAMA_GW_ABLATE bit 0 = activation loads answered by the buffer range check (no memory access), bit 1 = stores dropped
O=gpurun_out/${1:-w4abl}; mkdir -p $O
export LAMA_HIP_LIB=$PWD/lama_amd/lib/liblama_hip_prof.so
for abl in 0 1 2 3; do
  echo "== g4_trace fuconv LAMA_GW_ABLATE=$abl" | tee -a $O/summary.txt
  LAMA_GW_ABLATE=$abl LAMA_GEMM_W4=1 timeout 120 python tools/g4_trace.py fuconv 6 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt
done
