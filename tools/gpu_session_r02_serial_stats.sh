#!/bin/bash
# round 2: rocprofv3 --kernel-trace --stats of the bench in SERIAL launch order (every kernel alone on the GPU: the durations bench.py's roofline quotes)
O=gpurun_out/r02serial
mkdir -p $O
export TMPDIR=/tmp
(cd /tmp && LAMA_OVERLAP_STREAMS=0 timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-f32-leg --no-eager-leg > $GRAFT_REPO_ROOT/$O/prof_bench.log 2>&1)
tail -c 300 $O/prof_bench.log
for db in $(find $O/prof -name '*.db' | head -1); do python tools/rocpd_summary.py $db $O/kernel_stats.csv; done
rm -rf $O/prof
head -16 $O/kernel_stats.csv | cut -c1-170
