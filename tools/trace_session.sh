#!/bin/bash
# profiling session: per-workgroup timelines of the spectral-branch kernels
OUT=gpurun_out/${1:-trace}
mkdir -p $OUT
for cfg in "0 fuconv" "1 fuconv" "0 conv1" "1 conv1"; do
  set -- $cfg
  echo "== wr_trace LAMA_CW_1X1=$1 $2"
  LAMA_CW_1X1=$1 timeout 120 python tools/wr_trace.py $2 2>&1 | grep -v amdgpu.ids
done | tee $OUT/wr_trace.txt
for k in rfft irfft; do timeout 120 python tools/fft_trace.py $k 2>&1 | grep -v amdgpu.ids; done | tee $OUT/fft_trace.txt
