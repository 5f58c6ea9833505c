#!/bin/bash
# usage: mk_wk_bench.sh <tag> [extra hipcc flags] -> abtmp/wk_bench_<tag>
# builds it from lama_amd/csrc/gemm_wk_dev.inc + conv_ws_dev.inc (stand-alone harness of the spectral GEMM kernels); extra hipcc flags pass through
cd "$(dirname "$0")/../.."
TAG=${1:-base}; shift
mkdir -p abtmp/wk
( echo '#define CB_F16 1'; sed -n 1,208p lama_amd/csrc/conv_split3.inc; echo '#include "conv_ws_dev.inc"'; echo '#include "gemm_wk_dev.inc"'; echo '}'; cat tools/ubench/wk_bench_main.inc ) > abtmp/wk/wk_bench.hip
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-gpu-rdc -Iinclude -Ilama_amd/csrc -Xclang -target-feature -Xclang -packed-fp32-ops -save-temps=obj -Rpass-analysis=kernel-resource-usage "$@" abtmp/wk/wk_bench.hip -o abtmp/wk_bench_$TAG 2>&1 | grep -A9 "Name: .*gemm1x1_wk.*ILb0" | grep -E "VGPRs|AGPRs|Scratch|Spill|Occupancy"
ls -la abtmp/wk_bench_$TAG | awk '{print $5, $9}'
