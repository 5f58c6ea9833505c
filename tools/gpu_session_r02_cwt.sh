#!/bin/bash
# transposed convs as four parity-class launches of conv_wr_kernel (cwt_try_launch) against the fused LDS-staged launch: same-box A/B
# through the profiling build (LAMA_CWT=0/1), kernel table of both, then the parity tests that cover it
O=gpurun_out/r02cwt
mkdir -p $O
export LAMA_HIP_LIB=$PWD/lama_amd/lib/liblama_hip_prof.so
for f in 0 1 0 1; do echo -n "LAMA_CWT=$f " >> $O/ab.txt; LAMA_CWT=$f timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-f32-leg --no-eager-leg 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); k=d['kernels_us']; print(d['value'], d['ms_per_step'], {n:v for n,v in k.items() if 'T_' in n})" >> $O/ab.txt; done
cat $O/ab.txt
unset LAMA_HIP_LIB
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "conv2d" 2>&1 | tail -3 | tee $O/pytest_kernels.txt
