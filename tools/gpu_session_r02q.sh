#!/bin/bash
# round 2, session q: stride-2 convs, two 4-wave workgroups per CU vs one 8-wave workgroup vs the LDS-staged kernel
mkdir -p gpurun_out/r02q
O=gpurun_out/r02q
export LAMA_HIP_LIB=$PWD/lama_amd/lib/liblama_hip_prof.so
for w in "1 1" "1 2" "0 1" "1 1" "1 2" "0 1"; do set -- $w; echo -n "CONV_WR=$1 CW_S2=$2 " >> $O/ab_down.txt; LAMA_CONV_WR=$1 LAMA_CW_S2=$2 KPROBE_ITERS=30 python tools/kprobe.py f16x3 down1 down2 down3 2>&1 | grep down | tr '\n' ' ' >> $O/ab_down.txt; echo >> $O/ab_down.txt; done
cat $O/ab_down.txt
unset LAMA_HIP_LIB
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q > $O/pytest_kernels.log 2>&1; tail -3 $O/pytest_kernels.log
timeout 600 python bench.py --no-f32-leg --no-cpu-baseline --no-eager-leg > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02q/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('avg_us'), json.dumps(d.get('configs2_fp16_leg'))[:300])
PY
