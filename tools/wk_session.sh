#!/bin/bash
# same-box session of the spectral GEMM kernels: stand-alone harness (abtmp/wk_bench_<variant>, built by tools/ubench/mk_wk_bench.sh: wk with fifth waves /
# wk without / w4, plain and with the memory accesses ablated, timeline of the wk kernel), then the parity tests that reach the kernel, then bench
# A/B with the profiling library's switch.
# usage (through gpurun): bash tools/wk_session.sh <tag> [harness|tests|ab ...]   (default: all three)
TAG=${1:?tag}; shift; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
STEPS=${@:-harness tests ab}
for S in $STEPS; do case $S in
  harness) for f in abtmp/wk_bench_*; do [ -x $f ] || continue
             for w in 0 1; do timeout 60 $f 2112 $w 0; done; timeout 60 $f 2112 0 3
           done 2>&1 | grep -v amdgpu.ids | tee $O/harness.txt
           f=$(ls abtmp/wk_bench_* | head -1); timeout 60 $f 2112 2 0 2>&1 | grep -v amdgpu.ids | tee -a $O/harness.txt ;;
  tests)   timeout 1200 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "wks or big_tiles or fourier_unit or range_watch" > $O/pytest_wk.log 2>&1; tail -4 $O/pytest_wk.log ;;
  ab)      bash tools/session.sh $TAG "ab:LAMA_GEMM_WK=0,1" ;;
esac; done
