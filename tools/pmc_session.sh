#!/bin/bash
# rocprofv3 PMC passes over tools/kprobe.py.  usage: tools/pmc_session.sh <tag> "<kprobe args>" "<pass1 counters>" "<pass2 counters>" ...
TAG=$1; shift
ARGS=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
python $GRAFT_REPO_ROOT/tools/kprobe.py $ARGS > $OUT/plain.log 2>&1
cat $OUT/plain.log
i=0
for CNT in "$@"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $OUT/p$i -o pmc -- python $GRAFT_REPO_ROOT/tools/kprobe.py $ARGS > $OUT/p$i.log 2>&1
  echo "pass $i ($CNT): rc=$?"
  f=$(find $OUT/p$i -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python $GRAFT_REPO_ROOT/tools/pmc_summary.py $f | tee $OUT/p$i.summary.txt
  rm -rf $OUT/p$i
done
