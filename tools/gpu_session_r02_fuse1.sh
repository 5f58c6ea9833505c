#!/bin/bash
# round 2: SpectralTransform.conv1 of the next layer in the epilogue of the global-branch launch -- tests and A/B
O=gpurun_out/r02fuse1
mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "fused_next_conv1" 2>&1 | tail -4 | tee $O/summary.txt
timeout 900 python -m pytest tests/test_generator_gpu.py -m gpu -x -q 2>&1 | tail -4 | tee -a $O/summary.txt
for f in 1 0 1; do echo -n "LAMA_FUSE_CONV1=$f " >> $O/ab_fuse1.txt; LAMA_FUSE_CONV1=$f python bench.py --no-f32-leg --no-cpu-baseline --no-eager-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['kernels_us'])" | cut -c1-600 >> $O/ab_fuse1.txt; done
cat $O/ab_fuse1.txt
