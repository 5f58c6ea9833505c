#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r02b
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -q --maxfail=8 2>&1 | tail -40 ) > gpurun_out/r02b/pytest.log 2>&1
timeout 900 python tools/race_probe6.py 1500 conv:none:full conv:event:full conv:flush:full torch:none:full conv:none:fft conv:none:torchchain torch:none:torchchain none:none:full > gpurun_out/r02b/race6.log 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-eager-leg > gpurun_out/r02b/bench.json 2> gpurun_out/r02b/bench.err
tail -6 gpurun_out/r02b/pytest.log | cut -c1-300; cat gpurun_out/r02b/race6.log | cut -c1-400; python -c "
import json; d=json.load(open('gpurun_out/r02b/bench.json')); print(d['value'], d['ms_per_step'], d['kernels_us'])"
