#!/bin/bash
# round 2, session k: fp16 path with the fp32 residual stream -- kernel tests, generator test, bench with the configs[2] leg
mkdir -p gpurun_out/r02k
timeout 1500 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "fp16" > gpurun_out/r02k/pytest_fp16_kernels.log 2>&1
tail -5 gpurun_out/r02k/pytest_fp16_kernels.log
timeout 900 python -m pytest tests/test_generator_gpu.py -m gpu -q -k "fp16" > gpurun_out/r02k/pytest_fp16_gen.log 2>&1
tail -15 gpurun_out/r02k/pytest_fp16_gen.log
timeout 600 python bench.py > gpurun_out/r02k/bench.json 2> gpurun_out/r02k/bench.err
tail -c 3000 gpurun_out/r02k/bench.json
