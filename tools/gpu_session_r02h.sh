#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r02h
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -q --maxfail=5 -k "mfma_load or overlap_streams or range_watch or fourier_unit" -s 2>&1 | tail -6 ) > gpurun_out/r02h/pytest.log 2>&1
timeout 900 python tools/overlap_stress.py 100000 64 > gpurun_out/r02h/stress.log 2>&1
timeout 300 python tools/det_probe.py >> gpurun_out/r02h/stress.log 2>&1
timeout 600 python bench.py --steps 30 --warmup 5 > gpurun_out/r02h/bench.json 2> gpurun_out/r02h/bench.err
tail -6 gpurun_out/r02h/pytest.log | cut -c1-250; grep -E "==|overlap|runs" gpurun_out/r02h/stress.log | cut -c1-250
python -c "
import json; d=json.load(open('gpurun_out/r02h/bench.json')); print(d['value'], d['ms_per_step'], d['value_with_h2d_d2h'], d['pytorch_rocm_eager'], d['kernels_us'])"
