#!/bin/bash
# rocprofv3 counter passes (one --pmc set per run; kernel trace only) over tools/shape_probe.py: pmc_shape.sh <outdir> "<counters>" <shape>...
# prints per-kernel averages of the FFT kernels
OUT=$1; CNTS=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $OUT
export TMPDIR=/tmp
for C in $CNTS; do
  (cd /tmp && PROBE_STEPS=1 timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $ROOT/$OUT/p_$C -o pmc -- python $ROOT/tools/shape_probe.py "$@" > $ROOT/$OUT/p_$C.log 2>&1)
  f=$(find $OUT/p_$C -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python tools/pmc_summary.py $f | grep -E "mr2_|mr_kernel|fft" | cut -c1-70,118-200
  rm -rf $OUT/p_$C
done
