#!/bin/bash
# One gpurun session: same-box A/B of the spectral-branch kernels (pointwise GEMM kernel, FFT planes per workgroup).
# usage: tools/ab_fu.sh <tag>
TAG=${1:-abfu}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== tests" | tee $OUT/summary.txt
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "big_tiles or fft_sequential or fourier_unit or rfft2 or k1s1" 2>&1 | tail -6 | tee -a $OUT/summary.txt
for cfg in "0 0 1" "1 1 1"; do
  set -- $cfg
  echo "== kbench fu  LAMA_GEMM_WS=$1 LAMA_CW_1X1=$2 LAMA_FFT_SEQ=$3" | tee -a $OUT/summary.txt
  LAMA_GEMM_WS=$1 LAMA_CW_1X1=$2 LAMA_FFT_SEQ=$3 timeout 200 python tools/kbench.py f16x3 fu ws$1_cw$2_seq$3 > $OUT/kbench_ws$1_cw$2_seq$3.log 2>&1
  grep -E "^(conv1x1|rfft2|irfft2|fourier)" $OUT/kbench_ws$1_cw$2_seq$3.log | cut -c1-150 | tee -a $OUT/summary.txt
done
if [ "$2" != "nobench" ]; then
echo "== bench (new defaults)" | tee -a $OUT/summary.txt
timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-f32-leg > $OUT/bench.log 2>&1; tail -1 $OUT/bench.log | cut -c1-300 | tee -a $OUT/summary.txt
echo "== bench (old: LAMA_GEMM_WS=0 LAMA_CW_1X1=0 LAMA_FFT_SEQ=1)" | tee -a $OUT/summary.txt
LAMA_GEMM_WS=0 LAMA_CW_1X1=0 LAMA_FFT_SEQ=1 timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-f32-leg > $OUT/bench_old.log 2>&1; tail -1 $OUT/bench_old.log | cut -c1-300 | tee -a $OUT/summary.txt
fi
cp gpurun_out/kbench_*.json $OUT/ 2>/dev/null
echo "== done" | tee -a $OUT/summary.txt
