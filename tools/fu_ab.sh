#!/bin/bash
# same-box A/B of the FourierUnit routes (profiling library): LAMA_FU_SPLIT=0 fp32 spectrum, 1 pre-split spectrum (default)
O=gpurun_out/${1:-fuab}; mkdir -p $O
export LAMA_HIP_LIB=$PWD/lama_amd/lib/liblama_hip_prof.so KBENCH_ROT=6
for i in 1 2; do for v in 0 1; do
  echo "== kbench fu LAMA_FU_SPLIT=$v" | tee -a $O/summary.txt
  LAMA_FU_SPLIT=$v timeout 200 python tools/kbench.py f16x3 fu fusplit_$v 2>&1 | grep -E "fourier_unit" | cut -c1-150 | tee -a $O/summary.txt
done; done
