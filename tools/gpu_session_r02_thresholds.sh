#!/bin/bash
# round 2: where the cooperative local-conv geometry starts to pay (workgroups per launch), and the 4 x 1 two-per-CU geometry at >= 512 workgroups
O=gpurun_out/r02thr
mkdir -p $O
export LAMA_HIP_LIB=$PWD/lama_amd/lib/liblama_hip_prof.so
run() { echo -n "$1 x $2 LAMA_CW_41=$3 " >> $O/ab.txt; LAMA_CW_41=$3 LAMA_BENCH_BATCH=$1 LAMA_BENCH_RES=$2 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-f32-leg --no-eager-leg 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> $O/ab.txt; }
for cfg in "4 512" "6 512"; do set -- $cfg; for f in 0 -1 0 -1; do run $1 $2 $f; done; done
for cfg in "16 512" "4 1024"; do set -- $cfg; for f in 0 1 0 1; do run $1 $2 $f; done; done
cat $O/ab.txt
