#!/bin/bash
# usage: mk_head_bench.sh <tag> [extra hipcc flags] -> abtmp/head_bench_<tag> (stand-alone harness of head7_ws_kernel, tools/ubench/head_bench_main.inc)
cd "$(dirname "$0")/../.."
TAG=${1:-base}; shift
mkdir -p abtmp/head
( echo '#define CB_F16 1'; sed -n 1,208p lama_amd/csrc/conv_split3.inc; echo '#include "conv_head_dev.inc"'; echo '}'; cat tools/ubench/head_bench_main.inc ) > abtmp/head/head_bench.hip
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-gpu-rdc -Iinclude -Ilama_amd/csrc -Xclang -target-feature -Xclang -packed-fp32-ops -Rpass-analysis=kernel-resource-usage "$@" abtmp/head/head_bench.hip -o abtmp/head_bench_$TAG 2>&1 | grep -A9 "Name: .*head7_ws" | grep -E "VGPRs:|Scratch" | sed 's/remark: [^ ]* //' | tr '\n' ' '; echo " -> abtmp/head_bench_$TAG"
