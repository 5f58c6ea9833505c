#!/bin/bash
# same-box A/B of the weights-in-registers conv kernels: kernel parity tests, per-kernel timings with LAMA_CONV_WR=1/0, bench
TAG=${1:-wr}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== kernel tests" | tee $OUT/summary.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q 2>&1 | tail -8 | tee -a $OUT/summary.txt
for wr in 1 0 1 0; do
  echo "== kprobe LAMA_CONV_WR=$wr" | tee -a $OUT/summary.txt
  LAMA_CONV_WR=$wr KPROBE_ITERS=30 timeout 300 python tools/kprobe.py f16x3 convA convB conv1 fuconv 2>&1 | grep " us" | tr '\n' ' ' | tee -a $OUT/summary.txt
  echo | tee -a $OUT/summary.txt
done
for wr in 1 0; do
  echo "== bench LAMA_CONV_WR=$wr" | tee -a $OUT/summary.txt
  LAMA_CONV_WR=$wr timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-f32-leg > $OUT/bench_wr$wr.log 2>&1; tail -1 $OUT/bench_wr$wr.log | cut -c1-200 | tee -a $OUT/summary.txt
done
echo "== done" | tee -a $OUT/summary.txt
