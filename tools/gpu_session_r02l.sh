#!/bin/bash
# round 2, session l: row-rolling fragment reads of the weights-in-registers 3x3 kernel -- A/B, ablations, tests, bench
mkdir -p gpurun_out/r02l
O=gpurun_out/r02l
export LAMA_HIP_LIB=$PWD/lama_amd/lib/liblama_hip_prof.so
for r in 1 0 1 0; do echo -n "ROLL=$r " >> $O/ab_roll.txt; LAMA_CW_ROLL=$r KPROBE_ITERS=50 python tools/kprobe.py f16x3 convA 2>&1 | grep convA >> $O/ab_roll.txt; done
for a in 0 1 2 4 8 3 7 16 32; do echo -n "ROLL=0 ABL=$a " >> $O/ab_roll.txt; LAMA_CW_ROLL=0 LAMA_CW_ABLATE=$a KPROBE_ITERS=30 python tools/kprobe.py f16x3 convA 2>&1 | grep convA >> $O/ab_roll.txt; done
cat $O/ab_roll.txt
unset LAMA_HIP_LIB
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q > $O/pytest_kernels.log 2>&1; tail -3 $O/pytest_kernels.log
timeout 900 python -m pytest tests/test_generator_gpu.py -m gpu -x -q -k "fp16 or c2 or biglama" > $O/pytest_gen.log 2>&1; tail -5 $O/pytest_gen.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02l/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('avg_us'), json.dumps(d.get('configs2_fp16_leg'))[:300])
PY
