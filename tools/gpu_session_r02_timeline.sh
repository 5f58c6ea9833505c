#!/bin/bash
# round 2: per-dispatch timeline of one hipGraph step with the spectral branch on the second stream (tools/timeline.py)
O=gpurun_out/r02tl
mkdir -p $O
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$O/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-f32-leg --no-eager-leg > $GRAFT_REPO_ROOT/$O/bench.log 2>&1)
tail -c 600 $O/bench.log
for db in $(find $O/prof -name '*.db' | head -1); do python tools/timeline.py $db $O/timeline.txt 4; python tools/timeline.py $db $O/timeline_serial.txt 1; done
rm -rf $O/prof
sed -n 1,3p $O/timeline.txt | cut -c1-300; sed -n 40,75p $O/timeline.txt; tail -2 $O/timeline.txt
