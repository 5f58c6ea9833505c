#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r02f
export TMPDIR=/tmp
{
echo "#### per-opcode probes next to the MFMA spinner (profiling build)"
LAMA_HIP_LIB=lama_amd/lib/liblama_hip_prof.so timeout 600 python tools/race_probe9.py 100
echo "#### FFT kernels compiled with -target-feature -packed-fp32-ops (no v_pk_*_f32), same probes"
LAMA_HIP_LIB=lama_amd/lib/liblama_hip_nopk.so timeout 600 python tools/race_probe8.py 300
LAMA_HIP_LIB=lama_amd/lib/liblama_hip_nopk.so timeout 600 python tools/race_probe7.py 1000 fft_without_packed_fp32
} > gpurun_out/r02f/race.log 2>&1
grep -E "==|####" gpurun_out/r02f/race.log | grep -v "next to alone\|next to valu_hog" | cut -c1-330; tail -2 gpurun_out/r02f/race.log | cut -c1-200
