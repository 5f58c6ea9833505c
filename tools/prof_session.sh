#!/bin/bash
# rocprofv3 kernel stats of the bench workload (graph replay): per-kernel average durations inside the real pipeline
TAG=${1:-prof}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-f32-leg > $GRAFT_REPO_ROOT/$OUT/prof_bench.log 2>&1)
for db in $(find $OUT/prof -name '*.db' | head -1); do python tools/rocpd_summary.py $db $OUT/kernel_stats.csv; done
head -30 $OUT/kernel_stats.csv | cut -c1-200
tail -1 $OUT/prof_bench.log | cut -c1-200
rm -rf $OUT/prof
