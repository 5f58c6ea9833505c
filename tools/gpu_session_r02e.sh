#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r02e
export TMPDIR=/tmp
LAMA_HIP_LIB=lama_amd/lib/liblama_hip_prof.so timeout 600 python tools/race_probe9.py 150 > gpurun_out/r02e/race9.log 2>&1
grep "==" gpurun_out/r02e/race9.log | cut -c1-420; tail -3 gpurun_out/r02e/race9.log | cut -c1-300
