#!/bin/bash
# rocprofv3 kernel stats of the FourierUnit alone (tools/kprobe.py fu: 6 rotated operand sets), fp32-spectrum and pre-split routes
O=gpurun_out/${1:-fustats}; mkdir -p $O
ROOT=$PWD
export TMPDIR=/tmp LAMA_HIP_LIB=$PWD/lama_amd/lib/liblama_hip_prof.so KPROBE_ITERS=30
for v in ${2:-0 1}; do
  (cd /tmp && LAMA_FU_SPLIT=$v timeout 300 rocprofv3 --kernel-trace --stats -d $ROOT/$O/prof_$v -o fu -- python $ROOT/tools/kprobe.py f16x3 fu > $ROOT/$O/prof_$v.log 2>&1)
  for db in $(find $O/prof_$v -name '*.db' | head -1); do python tools/rocpd_summary.py $db $O/kernel_stats_fu_split$v.csv; done
  rm -rf $O/prof_$v
  echo "== FourierUnit kernels, LAMA_FU_SPLIT=$v" | tee -a $O/summary.txt
  grep -E "fft|gemm|conv" $O/kernel_stats_fu_split$v.csv | cut -c1-200 | head -8 | tee -a $O/summary.txt
done
