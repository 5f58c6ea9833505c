#!/bin/bash
# round 2, session m: timeline of the local 3x3 kernel, global branch 12x1 vs three 4x2 tiles with row-rolling reads
mkdir -p gpurun_out/r02m
O=gpurun_out/r02m
export LAMA_HIP_LIB=$PWD/lama_amd/lib/liblama_hip_prof.so
python tools/wr_trace.py convA > $O/wr_trace_convA.txt 2>&1; cat $O/wr_trace_convA.txt
for g in 1 0 1 0; do echo -n "G12=$g " >> $O/ab_g12.txt; LAMA_CW_G12=$g KPROBE_ITERS=50 python tools/kprobe.py f16x3 convB 2>&1 | grep convB >> $O/ab_g12.txt; done
for g in 1 0; do echo -n "bench G12=$g " >> $O/ab_g12.txt; LAMA_CW_G12=$g python bench.py --no-f32-leg --no-cpu-baseline --no-eager-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" >> $O/ab_g12.txt; done
cat $O/ab_g12.txt
