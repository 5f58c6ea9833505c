#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r02d
export TMPDIR=/tmp
LAMA_HIP_LIB=lama_amd/lib/liblama_hip_prof.so timeout 600 python tools/race_probe8.py 400 > gpurun_out/r02d/race8.log 2>&1
grep "==" gpurun_out/r02d/race8.log | cut -c1-300; tail -3 gpurun_out/r02d/race8.log | cut -c1-300
