#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r02i
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_refinement_gpu.py -m gpu -q --maxfail=5 2>&1 | tail -25 ) > gpurun_out/r02i/pytest.log 2>&1
tail -25 gpurun_out/r02i/pytest.log | cut -c1-300
