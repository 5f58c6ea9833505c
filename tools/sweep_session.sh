#!/bin/bash
# batch sweep at 512x512 (graph replay, quick bench) and rocprofv3 kernel stats of the LAMA_PREC_F16 generator at 4 x 1024^2
O=gpurun_out/${1:-sweep}; mkdir -p $O
ROOT=$PWD; export TMPDIR=/tmp
for b in 1 2 4 8 16; do
  LAMA_BENCH_BATCH=$b timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-f32-leg --no-eager-leg 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('batch $b x 512^2:', d['value'], 'images/s', d['ms_per_step'], 'ms per batch')" | tee -a $O/summary.txt
done
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $ROOT/$O/prof -o fp16 -- python $ROOT/tools/fp16_profile.py 14 > $ROOT/$O/prof_fp16.log 2>&1)
for db in $(find $O/prof -name '*.db' | head -1); do python tools/rocpd_summary.py $db $O/kernel_stats_fp16_4x1024.csv; done
rm -rf $O/prof; head -14 $O/kernel_stats_fp16_4x1024.csv | cut -c1-180 | tee -a $O/summary.txt
