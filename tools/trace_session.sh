#!/bin/bash
OUT=gpurun_out/${1:-trace}
mkdir -p $OUT
for a in 0 3; do echo "== LAMA_GW_ABLATE=$a"; LAMA_GW_ABLATE=$a timeout 120 python tools/gw_trace.py fuconv 6 2>&1 | grep -v amdgpu.ids | tail -9; done | tee $OUT/gw_abl.txt
timeout 120 python tools/gw_trace.py conv1 6 2>&1 | grep -v amdgpu.ids | tail -9 | tee -a $OUT/gw_abl.txt
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-f32-leg 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); print({k:v for k,v in list(d['kernels_us'].items())[:4]})" | tee -a $OUT/gw_abl.txt
