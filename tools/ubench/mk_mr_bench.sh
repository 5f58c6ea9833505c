#!/bin/bash
# usage: mk_mr_bench.sh <tag> [extra hipcc flags, e.g. -DMR_ABL=3] -> abtmp/mr_bench_<tag>
# stand-alone harness of the mixed-radix FFT kernels: fft.hip (with fft_mr_dev.inc) + tools/ubench/mr_bench_main.inc in one binary
cd "$(dirname "$0")/../.."
TAG=${1:-base}; shift
mkdir -p abtmp/mr
( echo '#include "fft.hip"'; cat tools/ubench/mr_bench_main.inc ) > abtmp/mr/mr_bench_$TAG.hip
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-gpu-rdc -Iinclude -Ilama_amd/csrc -Xclang -target-feature -Xclang -packed-fp32-ops -w "$@" abtmp/mr/mr_bench_$TAG.hip -o abtmp/mr_bench_$TAG 2>&1 | grep -v "recognized feature" | head; echo " -> abtmp/mr_bench_$TAG"
