#!/bin/bash
# round 2, after the cooperative overlap: what the second stream and the fused conv1 are worth NOW (same box, product library)
O=gpurun_out/r02abend
mkdir -p $O
run() { echo -n "$1 " >> $O/ab.txt; env $1 python bench.py --no-f32-leg --no-cpu-baseline --no-eager-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" >> $O/ab.txt; }
for v in 1 0 1 0; do run LAMA_OVERLAP_STREAMS=$v; done
for v in 1 0 1 0; do run LAMA_FUSE_CONV1=$v; done
cat $O/ab.txt
