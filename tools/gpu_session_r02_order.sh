#!/bin/bash
# round 2: capture order at the fork (local conv first = it stays on the global launch's queue inside the hipGraph) -- A/B + timeline
O=gpurun_out/r02order
mkdir -p $O
export TMPDIR=/tmp
for f in 1 0 1 0; do echo -n "LAMA_LOCAL_FIRST=$f " >> $O/ab_order.txt; LAMA_LOCAL_FIRST=$f python bench.py --no-f32-leg --no-cpu-baseline --no-eager-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" >> $O/ab_order.txt; done
cat $O/ab_order.txt
(cd /tmp && LAMA_LOCAL_FIRST=1 timeout 600 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$O/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-f32-leg --no-eager-leg > $GRAFT_REPO_ROOT/$O/bench.log 2>&1)
for db in $(find $O/prof -name '*.db' | head -1); do python tools/timeline.py $db $O/timeline.txt 4; done
rm -rf $O/prof
sed -n 40,56p $O/timeline.txt; tail -1 $O/timeline.txt
