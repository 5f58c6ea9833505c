#!/bin/bash
# same-box A/B of the pointwise GEMM kernels (profiling library): LAMA_GEMM_W4=0 (32-pixel tiles) vs 1 (64-pixel super-tiles)
O=gpurun_out/${1:-w4}; mkdir -p $O
export LAMA_HIP_LIB=$PWD/lama_amd/lib/liblama_hip_prof.so KBENCH_ROT=6
for i in 1 2; do for v in 0 1; do
  echo "== kbench fu LAMA_GEMM_W4=$v" | tee -a $O/summary.txt
  LAMA_GEMM_W4=$v timeout 200 python tools/kbench.py f16x3 fu w4_$v 2>&1 | grep -E "conv1x1|fourier_unit|rfft2|irfft2" | cut -c1-150 | tee -a $O/summary.txt
done; done
for n in fuconv; do
  echo "== g4_trace $n" | tee -a $O/summary.txt
  LAMA_GEMM_W4=1 timeout 120 python tools/g4_trace.py $n 6 2>&1 | tail -12 | tee -a $O/summary.txt
done
