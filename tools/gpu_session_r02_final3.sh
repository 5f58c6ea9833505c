#!/bin/bash
# round 2, closing session (cooperative local conv, wave priority, capture order in): full GPU suite, smoke, default bench, rocprofv3 stats +
# per-dispatch timeline of the same command, other shapes, small-launch A/B of the cooperative geometry
O=gpurun_out/r02final3
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log | tee $O/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a $O/summary.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 400 $O/bench.err
python - <<'PY' | tee -a gpurun_out/r02final3/summary.txt
import json
d=json.loads(open('gpurun_out/r02final3/bench.json').read().strip().splitlines()[-1])
r=d['roofline']
print('bench:', d['value'], d['unit'], d['ms_per_step'], 'ms; roofline', r['kernel'], r['achieved'], 'TF frac', r['frac'], 'sustained', (r.get('peak_sustained') or {}).get('value'), 'frac_of_sustained', r.get('frac_of_sustained'), 'runner_up', r.get('runner_up'))
print('roofline_ffc:', d['roofline_ffc']['avg_us'], 'us frac', d['roofline_ffc']['frac'])
print('eager:', (d.get('pytorch_rocm_eager') or {}).get('value'), 'pcie:', (d.get('value_with_h2d_d2h') or {}).get('value'), 'cpu:', d['cpu_baseline']['value'], 'f32:', (d.get('exact_f32_leg') or {}).get('value'))
print('configs2:', json.dumps(d.get('configs2_fp16_leg'))[:260])
print('configs4:', json.dumps(d.get('configs4_refine_leg'))[:200])
print('kernels_us:', json.dumps(d.get('kernels_us'))[:900])
PY
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-f32-leg --no-eager-leg > $GRAFT_REPO_ROOT/$O/prof_bench.log 2>&1)
for db in $(find $O/prof -name '*.db' | head -1); do python tools/rocpd_summary.py $db $O/kernel_stats.csv; python tools/timeline.py $db $O/timeline.txt 4; done
rm -rf $O/prof
head -12 $O/kernel_stats.csv | cut -c1-170
sed -n 40,52p $O/timeline.txt; tail -1 $O/timeline.txt
for cfg in "4 1024" "4 256" "1 512" "1 2048"; do set -- $cfg
  LAMA_BENCH_BATCH=$1 LAMA_BENCH_RES=$2 timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-f32-leg --no-eager-leg 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$1 x $2:', d['value'], 'images/s', d['ms_per_step'], 'ms')" | tee -a $O/summary.txt
done
export LAMA_HIP_LIB=$PWD/lama_amd/lib/liblama_hip_prof.so
for cfg in "4 256" "1 512" "2 512" "16 512"; do set -- $cfg
  for f in 0 -1 0 -1; do echo -n "$1 x $2 LAMA_CW_41=$f " >> $O/ab_coop_small.txt; LAMA_CW_41=$f LAMA_BENCH_BATCH=$1 LAMA_BENCH_RES=$2 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-f32-leg --no-eager-leg 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> $O/ab_coop_small.txt; done
done
cat $O/ab_coop_small.txt
