#!/bin/bash
# what do the step-2 stores of the parity-class launches cost?  LAMA_CWT=3 (profiling build, results wrong): the same four launches with
# contiguous stores
O=gpurun_out/r02cwt2
mkdir -p $O
export LAMA_HIP_LIB=$PWD/lama_amd/lib/liblama_hip_prof.so
for f in 0 1 3 0 1 3; do echo -n "LAMA_CWT=$f " >> $O/ab.txt; LAMA_CWT=$f timeout 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-f32-leg --no-eager-leg 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); k=d['kernels_us']; print(d['value'], d['ms_per_step'], {n:v for n,v in k.items() if 'T_' in n})" >> $O/ab.txt; done
cat $O/ab.txt
