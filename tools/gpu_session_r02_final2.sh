#!/bin/bash
# round 2, closing session after the bench fix (stream overlap on in the timed region): default bench + rocprofv3 kernel stats of it
O=gpurun_out/r02final2
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
python - <<'PY' | tee gpurun_out/r02final2/summary.txt
import json
d=json.loads(open('gpurun_out/r02final2/bench.json').read().strip().splitlines()[-1])
r=d['roofline']
print('bench:', d['value'], d['unit'], d['ms_per_step'], 'ms; roofline', r['kernel'], r['achieved'], 'TF frac', r['frac'], 'sustained', (r.get('peak_sustained') or {}).get('value'), 'frac_of_sustained', r.get('frac_of_sustained'), 'runner_up', r.get('runner_up'))
print('roofline_ffc:', d['roofline_ffc']['avg_us'], 'us frac', d['roofline_ffc']['frac'])
print('eager:', (d.get('pytorch_rocm_eager') or {}).get('value'), 'pcie:', (d.get('value_with_h2d_d2h') or {}).get('value'), 'cpu:', d['cpu_baseline']['value'], 'f32:', (d.get('exact_f32_leg') or {}).get('value'))
print('configs2:', json.dumps(d.get('configs2_fp16_leg'))[:260])
print('configs4:', json.dumps(d.get('configs4_refine_leg'))[:200])
PY
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-f32-leg --no-eager-leg > $GRAFT_REPO_ROOT/$O/prof_bench.log 2>&1)
for db in $(find $O/prof -name '*.db' | head -1); do python tools/rocpd_summary.py $db $O/kernel_stats.csv; done
rm -rf $O/prof
head -14 $O/kernel_stats.csv | cut -c1-170
