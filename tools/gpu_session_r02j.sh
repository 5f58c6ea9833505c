#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r02j
export TMPDIR=/tmp
( timeout 1800 python -m pytest tests -m gpu -q --maxfail=10 2>&1 | tail -30 ) > gpurun_out/r02j/pytest.log 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r02j/bench.json 2> gpurun_out/r02j/bench.err
tail -14 gpurun_out/r02j/pytest.log | cut -c1-250
python -c "
import json; d=json.load(open('gpurun_out/r02j/bench.json')); print(d['value'], d['ms_per_step'], d['configs2_fp16_leg'], d['kernels_us'])"
tail -3 gpurun_out/r02j/bench.err
