#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r02g
export TMPDIR=/tmp
{
echo "#### asm probes (profiling build, FFT TU now without packed fp32; the debug probes keep theirs)"
LAMA_HIP_LIB=lama_amd/lib/liblama_hip_prof.so timeout 300 python tools/race_probe9.py 60 2>&1 | grep "mfma_hog" | grep -v "^== rfft2"
echo "#### product library: overlapped chain vs serial"
timeout 300 python tools/race_probe7.py 3000 product_nopk
timeout 300 python tools/race_probe5.py 3000 32 full
timeout 300 python tools/race_probe5.py 3000 64 full
echo "#### generator level, overlap_streams on/off, eager/graph"
timeout 300 python tools/det_probe.py
} > gpurun_out/r02g/race.log 2>&1
( timeout 900 python -m pytest tests -m gpu -q --maxfail=5 2>&1 | tail -4 ) > gpurun_out/r02g/pytest.log 2>&1
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-eager-leg --no-f32-leg > gpurun_out/r02g/bench_prod.json 2> gpurun_out/r02g/bench_prod.err
LAMA_HIP_LIB=lama_amd/lib/liblama_hip_allnopk.so timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-eager-leg --no-f32-leg > gpurun_out/r02g/bench_allnopk.json 2> gpurun_out/r02g/bench_allnopk.err
LAMA_OVERLAP_STREAMS=1 timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-eager-leg --no-f32-leg > gpurun_out/r02g/bench_prod_overlap.json 2> gpurun_out/r02g/bench_prod_overlap.err
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-eager-leg --no-f32-leg > gpurun_out/r02g/bench_prod2.json 2> gpurun_out/r02g/bench_prod2.err
grep -E "==|####|overlap" gpurun_out/r02g/race.log | cut -c1-300; tail -3 gpurun_out/r02g/pytest.log
for f in prod allnopk prod_overlap prod2; do python -c "
import json,sys; d=json.load(open('gpurun_out/r02g/bench_$f.json')); print('$f', d['value'], d['ms_per_step'], {k:v for k,v in list(d['kernels_us'].items())[:4]})"; done
