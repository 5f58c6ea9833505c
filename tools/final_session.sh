#!/bin/bash
# Round-end session: parity tests, smoke, headline bench (with cpu_baseline + exact-f32 leg), rocprofv3 kernel stats, per-kernel bench
# with operands rotated out of the Infinity Cache, exploratory shapes.  usage: tools/final_session.sh <tag>
TAG=${1:-final}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest -m gpu" | tee $OUT/summary.txt
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee -a $OUT/summary.txt
echo "== smoke" | tee -a $OUT/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee -a $OUT/summary.txt
echo "== bench (default flags)" | tee -a $OUT/summary.txt
timeout 900 python bench.py > $OUT/bench.log 2>&1; tail -1 $OUT/bench.log > $OUT/bench.json; cut -c1-260 $OUT/bench.json | tee -a $OUT/summary.txt
echo "== rocprofv3 kernel stats of bench.py" | tee -a $OUT/summary.txt
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-f32-leg > $GRAFT_REPO_ROOT/$OUT/prof_bench.log 2>&1)
for db in $(find $OUT/prof -name '*.db' | head -1); do python tools/rocpd_summary.py $db $OUT/kernel_stats.csv; done
head -14 $OUT/kernel_stats.csv | cut -c1-150 | tee -a $OUT/summary.txt
rm -rf $OUT/prof
echo "== kbench (KBENCH_ROT=6)" | tee -a $OUT/summary.txt
KBENCH_ROT=6 timeout 300 python tools/kbench.py f16x3 all cold > $OUT/kbench_cold.log 2>&1; cp gpurun_out/kbench_cold.json $OUT/ 2>/dev/null; grep -c us $OUT/kbench_cold.log | tee -a $OUT/summary.txt
echo "== other shapes" | tee -a $OUT/summary.txt
for cfg in "4 1024" "4 256" "1 512" "1 2048"; do set -- $cfg
  LAMA_BENCH_BATCH=$1 LAMA_BENCH_RES=$2 timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-f32-leg 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$1 x $2:', d['value'], 'images/s', d['ms_per_step'], 'ms')" | tee -a $OUT/summary.txt
done
echo "== done" | tee -a $OUT/summary.txt
