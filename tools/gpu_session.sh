#!/bin/bash
# One gpurun session: parity tests, per-kernel bench, end-to-end bench, rocprofv3 kernel stats.
# usage: tools/gpu_session.sh <tag> [quick]
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== kernel tests" | tee $OUT/summary.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q 2>&1 | tail -15 | tee -a $OUT/summary.txt
echo "== kbench f16x3" | tee -a $OUT/summary.txt
timeout 300 python tools/kbench.py f16x3 > $OUT/kbench_f16x3.log 2>&1; tail -3 $OUT/kbench_f16x3.log | tee -a $OUT/summary.txt
echo "== bench" | tee -a $OUT/summary.txt
timeout 600 python bench.py --steps 20 --warmup 3 > $OUT/bench.log 2>&1; tail -2 $OUT/bench.log | tee -a $OUT/summary.txt
if [ "$2" != "quick" ]; then
  echo "== generator tests" | tee -a $OUT/summary.txt
  timeout 1500 python -m pytest tests/test_generator_gpu.py -m gpu -x -q 2>&1 | tail -15 | tee -a $OUT/summary.txt
  echo "== rocprofv3 kernel stats of bench.py" | tee -a $OUT/summary.txt
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-f32-leg > $GRAFT_REPO_ROOT/$OUT/prof_bench.log 2>&1)
  ls -R $OUT/prof | head -20 | tee -a $OUT/summary.txt
  for db in $(find $OUT/prof -name '*.db' | head -1); do python tools/rocpd_summary.py $db $OUT/kernel_stats.csv; done
  head -12 $OUT/kernel_stats.csv | cut -c1-160 | tee -a $OUT/summary.txt
  rm -rf $OUT/prof/*/*.db 2>/dev/null
fi
cp gpurun_out/kbench_*.json $OUT/ 2>/dev/null
echo "== done" | tee -a $OUT/summary.txt
