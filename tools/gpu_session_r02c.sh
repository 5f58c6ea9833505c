#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r02c
export TMPDIR=/tmp
P=lama_amd/lib/liblama_hip_prof.so
{
timeout 300 python tools/race_probe7.py 1000 product
LAMA_HIP_LIB=$P timeout 300 python tools/race_probe7.py 1000 prof_default
LAMA_HIP_LIB=$P LAMA_CONV_WR=0 timeout 300 python tools/race_probe7.py 1000 main_conv_lds_staged
LAMA_HIP_LIB=$P LAMA_FFT_INPLACE=0 timeout 300 python tools/race_probe7.py 1000 fft_two_buffer
LAMA_HIP_LIB=$P LAMA_GEMM_WS=0 timeout 300 python tools/race_probe7.py 1000 gemm_per_tile
} > gpurun_out/r02c/race7.log 2>&1
cat gpurun_out/r02c/race7.log | grep "==" | cut -c1-500
