#!/bin/bash
# batch 1 at 512^2: hipGraph replay against plain (eager) launches -- is a small step launch-bound or bound by one workgroup's K loop?
O=gpurun_out/r02small
mkdir -p $O
for g in "" "--no-graph" "" "--no-graph"; do echo -n "1 x 512 ${g:-graph} " >> $O/small.txt; LAMA_BENCH_BATCH=1 LAMA_BENCH_RES=512 timeout 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-f32-leg --no-eager-leg $g 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); k=d['kernels_us']; print(d['value'], 'images/s', d['ms_per_step'], 'ms', {n:v for n,v in list(k.items())[:3]})" >> $O/small.txt; done
cat $O/small.txt
