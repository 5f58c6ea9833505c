#!/bin/bash
# round 2, closing session: full GPU suite, smoke, default bench (the line the driver will reproduce), exploratory shapes
O=gpurun_out/r02final
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log | tee $O/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a $O/summary.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 400 $O/bench.err
python - <<'PY' | tee -a gpurun_out/r02final/summary.txt
import json
d=json.loads(open('gpurun_out/r02final/bench.json').read().strip().splitlines()[-1])
r=d['roofline']
print('bench:', d['value'], d['unit'], d['ms_per_step'], 'ms; roofline', r['kernel'], r['achieved'], 'TF frac', r['frac'], 'sustained', (r.get('peak_sustained') or {}).get('value'), 'frac_of_sustained', r.get('frac_of_sustained'))
print('roofline_ffc:', d['roofline_ffc']['avg_us'], 'us frac', d['roofline_ffc']['frac'])
print('eager:', (d.get('pytorch_rocm_eager') or {}).get('value'), 'pcie:', (d.get('value_with_h2d_d2h') or {}).get('value'), 'cpu:', d['cpu_baseline']['value'])
print('configs2:', json.dumps(d.get('configs2_fp16_leg'))[:260])
print('configs4:', json.dumps(d.get('configs4_refine_leg'))[:300])
PY
for cfg in "4 1024" "4 256" "1 512" "1 2048"; do set -- $cfg
  LAMA_BENCH_BATCH=$1 LAMA_BENCH_RES=$2 timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-f32-leg --no-eager-leg 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$1 x $2:', d['value'], 'images/s', d['ms_per_step'], 'ms')" | tee -a $O/summary.txt
done
