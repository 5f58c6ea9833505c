#!/bin/bash
# round 2, session p: stride-2 downsampling convs on the weights-in-registers kernel -- A/B, tests, bench; MFMA power micro-benchmark
mkdir -p gpurun_out/r02p
O=gpurun_out/r02p
hipcc --offload-arch=gfx950 -O3 -w tools/ubench/mfma_power.hip -o /tmp/mfma_power && /tmp/mfma_power > $O/mfma_power.txt 2>&1
export LAMA_HIP_LIB=$PWD/lama_amd/lib/liblama_hip_prof.so
for w in 1 0 1 0; do echo -n "CONV_WR=$w " >> $O/ab_down.txt; LAMA_CONV_WR=$w KPROBE_ITERS=30 python tools/kprobe.py f16x3 down1 down2 down3 2>&1 | grep down | tr '\n' ' ' >> $O/ab_down.txt; echo >> $O/ab_down.txt; done
cat $O/ab_down.txt
unset LAMA_HIP_LIB
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q > $O/pytest_kernels.log 2>&1; tail -3 $O/pytest_kernels.log
timeout 900 python -m pytest tests/test_generator_gpu.py -m gpu -x -q > $O/pytest_gen.log 2>&1; tail -5 $O/pytest_gen.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02p/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('avg_us'), json.dumps(d.get('configs2_fp16_leg'))[:300])
PY
