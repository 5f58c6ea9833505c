#!/bin/bash
# round 2: LAMA_CONV_COOPERATIVE (the local 3x3 conv as one 4-wave workgroup per CU beside the spectral branch) -- tests and A/B
O=gpurun_out/r02coop
mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "cooperative or overlap_streams or fused_next_conv1 or big_tiles" 2>&1 | tail -4 | tee $O/summary.txt
timeout 900 python -m pytest tests/test_generator_gpu.py -m gpu -x -q 2>&1 | tail -4 | tee -a $O/summary.txt
export LAMA_HIP_LIB=$PWD/lama_amd/lib/liblama_hip_prof.so
for f in 0 -1 0 -1; do echo -n "LAMA_CW_41=$f " >> $O/ab_coop.txt; LAMA_CW_41=$f python bench.py --no-f32-leg --no-cpu-baseline --no-eager-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['kernels_us'])" | cut -c1-400 >> $O/ab_coop.txt; done
unset LAMA_HIP_LIB
cat $O/ab_coop.txt
python bench.py > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json
