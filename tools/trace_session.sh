#!/bin/bash
OUT=gpurun_out/${1:-trace}
mkdir -p $OUT
for nh in 2 1; do
echo "== LAMA_GEMM_WS_NH=$nh"
for k in fuconv conv1; do LAMA_GEMM_WS_NH=$nh timeout 120 python tools/gw_trace.py $k 6 2>&1 | grep -v amdgpu.ids | grep -E "nrot|tile [0-9]:|prologue|total|thread 0"; done
LAMA_GEMM_WS_NH=$nh timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-f32-leg 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); print({k:v for k,v in list(d['kernels_us'].items())[:4]})"
done | tee $OUT/gw_nh.txt
