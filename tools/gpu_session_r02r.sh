#!/bin/bash
# round 2, session r: stride-1 3x3 at 128 x 128 planes, two 4-wave workgroups per CU vs one 8-wave K-split workgroup
mkdir -p gpurun_out/r02r
O=gpurun_out/r02r
export LAMA_HIP_LIB=$PWD/lama_amd/lib/liblama_hip_prof.so
for w in 0 1 0 1; do echo -n "CW_41=$w " >> $O/ab_41.txt; LAMA_CW_41=$w KPROBE_ITERS=20 python tools/kprobe.py f16x3 convA128 convA 2>&1 | grep conv | tr '\n' ' ' >> $O/ab_41.txt; echo >> $O/ab_41.txt; done
for w in 0 1; do echo -n "CW_41=$w 4x1024 " >> $O/ab_41.txt; LAMA_CW_41=$w LAMA_BENCH_BATCH=4 LAMA_BENCH_RES=1024 python bench.py --steps 8 --no-f32-leg --no-cpu-baseline --no-eager-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" >> $O/ab_41.txt; done
cat $O/ab_41.txt
