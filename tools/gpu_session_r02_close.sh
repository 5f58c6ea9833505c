#!/bin/bash
# round 2, last session: HEAD as the driver will run it -- full GPU suite, smoke, default bench, rocprofv3 kernel stats of the default command
O=gpurun_out/r02close
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log | tee $O/summary.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a $O/summary.txt
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err
python - <<'PY' | tee -a gpurun_out/r02close/summary.txt
import json
d=json.loads(open('gpurun_out/r02close/bench.json').read().strip().splitlines()[-1])
r=d['roofline']
print('bench:', d['value'], d['unit'], d['ms_per_step'], 'ms; roofline', r['kernel'], r['achieved'], 'TF frac', r['frac'], 'frac_of_sustained', r.get('frac_of_sustained'))
print('roofline_ffc:', d['roofline_ffc']['avg_us'], 'us frac', d['roofline_ffc']['frac'])
print('eager:', (d.get('pytorch_rocm_eager') or {}).get('value'), 'pcie:', (d.get('value_with_h2d_d2h') or {}).get('value'), 'cpu:', d['cpu_baseline']['value'])
PY
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-f32-leg --no-eager-leg > $GRAFT_REPO_ROOT/$O/prof_bench.log 2>&1)
for db in $(find $O/prof -name '*.db' | head -1); do python tools/rocpd_summary.py $db $O/kernel_stats.csv; done
rm -rf $O/prof
head -10 $O/kernel_stats.csv | cut -c1-170
