#!/bin/bash
# usage: mk_ct_bench.sh <tag> [extra hipcc flags, e.g. -DC2_ABL=1 -DC2_AR=3] -> abtmp/ct_bench_<tag>
# stand-alone harness of the ConvTranspose2d kernels (lama_amd/csrc/convt_dev.inc + tools/ubench/ct_bench_main.inc): 80 s to build, seconds to run
cd "$(dirname "$0")/../.."
TAG=${1:-base}; shift
mkdir -p abtmp/ct
( echo '#define CB_F16 1'; sed -n 1,208p lama_amd/csrc/conv_split3.inc; echo '#include "convt_dev.inc"'; echo '}'; cat tools/ubench/ct_bench_main.inc ) > abtmp/ct/ct_bench_$TAG.hip
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-gpu-rdc -Iinclude -Ilama_amd/csrc -Xclang -target-feature -Xclang -packed-fp32-ops -Rpass-analysis=kernel-resource-usage "$@" abtmp/ct/ct_bench_$TAG.hip -o abtmp/ct_bench_$TAG 2>&1 | grep -A9 "Name: .*convt2" | grep -E "VGPRs:|Scratch|VGPRs Spill" | sed 's/remark: [^ ]* //' | tr '\n' ' '; echo " -> abtmp/ct_bench_$TAG"
