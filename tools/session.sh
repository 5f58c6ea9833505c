#!/bin/bash
# ONE GPU-box session script (replaces the 38 one-off tools/gpu_session_r02*.sh of round 2; those are in the git history).
# Run through gpurun from the repo root:   gpurun --timeout 900 -- 'bash tools/session.sh <tag> <step> [<step> ...]'
# Results go to gpurun_out/<tag>/ (summary.txt collects one-line results); copy what is to be judged into profiles/.
#
# steps (run in the order given):
#   tests[:<pytest -k expr>]   pytest -m gpu (optionally -k expr)                          -> pytest_gpu.log
#   testsall                   pytest -m gpu without -x (every failure listed)
#   smoke                      __graft_entry__.smoke()
#   bench[:<extra flags>]      python bench.py <flags>                                     -> bench.json (+ one-line digest)
#   quick                      bench without the cpu / f32 / eager / configs legs          -> bench_quick.json
#   stats                      rocprofv3 --kernel-trace --stats of the quick bench         -> kernel_stats.csv, timeline.txt
#   overlap                    rocprofv3 kernel trace of tools/split_trace.py: which kernels ran side by side (one-part vs split plan) -> overlap_summary.txt
#   pmc:<kprobe names>         rocprofv3 --pmc passes (one counter set per run) over tools/kprobe.py f16x3 <names>  -> pmc_<names>/
#   pmcbench                   FETCH_SIZE / WRITE_SIZE passes over the quick bench itself (traffic IN the pipeline; under the profiler the generator keeps its one-part plan: verify_split) -> pmc_bench_*.txt, pmc.json
#   mfmabench                  SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE passes over the quick bench (run `stats` first: kernel_stats.csv gives the durations) -> mfma_util.txt
#   shapes                     the other single-GPU shapes (4x1024, 4x256, 1x512, 1x2048)
#   kbench                     per-kernel timings, operands rotated out of the Infinity Cache (KBENCH_ROT=6)
#   power                      MFMA sustained-rate micro-benchmark (tools/ubench/mfma_power.hip)
#   ab:<ENV>=<v1>,<v2>[,...]   same-box A/B of an environment switch of the PROFILING library, two rounds   -> ab_<ENV>.txt
#   ablib:<path/base.so>       same-box A/B of another build of the library against the in-tree one (bench.py --lib)
#   sh:<command>               bash -c <command>  (quote the step)                         -> sh_<hash>.log
#   py:<script> [args]         python <script> args  (quote the step)                      -> py_<script>.log
#   statspy:<script> [args]    rocprofv3 kernel stats of python <script> args              -> kernel_stats_<script>.csv
TAG=${1:?usage: session.sh <tag> <step>...}; shift
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$PWD}
QUICK="--steps 20 --warmup 3 --no-cpu-baseline --no-f32-leg --no-eager-leg"
digest() { python - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r, f = d.get('roofline') or {}, d.get('roofline_ffc') or {}
print('bench:', d['value'], d['unit'], d['ms_per_step'], 'ms | roofline', r.get('kernel'), r.get('achieved'), 'TF frac', r.get('frac'),
      '| ffc', f.get('avg_us'), 'us frac', f.get('frac'), 'traffic', f.get('traffic'))
for k in ('pytorch_rocm_eager', 'exact_f32_leg', 'cpu_baseline', 'configs2_fp16_leg', 'configs4_refine_leg', 'configs4_refine_default_leg', 'photo_leg', 'predict_cli_leg', 'value_host_fed', 'batch16_leg'):
    if d.get(k):
        print(' ', k, json.dumps(d[k])[:1500])
print('  kernels_us:', json.dumps(d.get('kernels_us'))[:1200])
PY
}
for STEP in "$@"; do
  KIND=${STEP%%:*}; ARG=""; [ "$KIND" != "$STEP" ] && ARG=${STEP#*:}
  echo "== $STEP" | tee -a $O/summary.txt
  case $KIND in
    tests)  if [ -n "$ARG" ]; then timeout 1500 python -m pytest tests -m gpu -x -q -k "$ARG" > $O/pytest_gpu.log 2>&1; else timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; fi
            tail -5 $O/pytest_gpu.log | tee -a $O/summary.txt ;;
    testsall) timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; grep -E "^(FAILED|ERROR)" $O/pytest_gpu.log | head -40 | tee -a $O/summary.txt; tail -3 $O/pytest_gpu.log | tee -a $O/summary.txt ;;
    smoke)  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a $O/summary.txt ;;
    bench)  timeout 1200 python bench.py $ARG > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err; digest $O/bench.json | tee -a $O/summary.txt ;;
    quick)  timeout 400 python bench.py $QUICK > $O/bench_quick.json 2> $O/bench_quick.err; tail -c 300 $O/bench_quick.err; digest $O/bench_quick.json | tee -a $O/summary.txt ;;
    stats)  # (round 6: --split-batch 1 --no-host-fed-leg: every launch in the trace is a whole-batch launch of the ONE-part plan: 2 warm-up + 10 timed graph replays,
            #  then 2 x (1 warm-up + 3 instrumented) eager steps = 20 steps, i.e. 35 x 20 = 700 calls of the fused global launch)
            # rocprofv3's kernel trace SERIALISES the queues a split plan runs on (profiles/r05_overlap_under_rocprof.txt: 33 ms per replay instead of
            # 9.4): the generator's own check (verify_split) sees that and keeps the one-part plan -- the same kernels over the whole batch
            (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $ROOT/$O/prof -o bench -- python $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-f32-leg --no-eager-leg --no-host-fed-leg --split-batch 1 > $ROOT/$O/prof_bench.log 2>&1)
            grep -o '"value": [0-9.]*, "unit": "images/s", "n_gpus"' $O/prof_bench.log | head -1 | tee -a $O/summary.txt; grep -o '"ms_per_step": [0-9.]*' $O/prof_bench.log | head -1 | tee -a $O/summary.txt; grep -o '"split_batch": [^}]*}' $O/prof_bench.log | head -1 | tee -a $O/summary.txt
            for db in $(find $O/prof -name '*.db' | head -1); do python tools/rocpd_summary.py $db $O/kernel_stats.csv; python tools/timeline.py $db $O/timeline.txt 0; done
            rm -rf $O/prof; head -16 $O/kernel_stats.csv | cut -c1-170 | tee -a $O/summary.txt ;;
    statspy) set -- $ARG; N=$(basename $1 .py)
            (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $ROOT/$O/prof_$N -o run -- python $ROOT/$1 ${@:2} > $ROOT/$O/prof_$N.log 2>&1)
            for db in $(find $O/prof_$N -name '*.db' | head -1); do python tools/rocpd_summary.py $db $O/kernel_stats_$N.csv; done
            rm -rf $O/prof_$N; tail -3 $O/prof_$N.log | tee -a $O/summary.txt; head -40 $O/kernel_stats_$N.csv | cut -c1-200 | tee -a $O/summary.txt ;;
    overlap) (cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $ROOT/$O/prof_split -o run -- python $ROOT/tools/split_trace.py 6 > $ROOT/$O/prof_split.log 2>&1)
            for db in $(find $O/prof_split -name '*.db' | head -1); do python tools/overlap_summary.py $db 6 | tee $O/overlap_summary.txt | tee -a $O/summary.txt; done
            rm -rf $O/prof_split; tail -2 $O/prof_split.log | tee -a $O/summary.txt ;;
    pmc)    bash tools/pmc_session.sh $TAG/pmc_$(echo $ARG | tr ' ' '_') "f16x3 $ARG" "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT" 2>&1 | tail -40 | tee -a $O/summary.txt ;;
    pmcbench) for CNT in FETCH_SIZE WRITE_SIZE; do
              (cd /tmp && timeout 600 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $ROOT/$O/pmcb_$CNT -o pmc -- python $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-f32-leg --no-eager-leg --no-host-fed-leg --split-batch 1 > $ROOT/$O/pmcb_$CNT.log 2>&1)
              f=$(find $O/pmcb_$CNT -name '*counter_collection.csv' | head -1); [ -n "$f" ] && python tools/pmc_summary.py $f > $O/pmc_bench_$CNT.txt && grep -E "conv_wr_kernel|wino_|gemm1x1_wk|fft2_ip64|convt2|head7|stem7" $O/pmc_bench_$CNT.txt | cut -c1-60,118-200 | tee -a $O/summary.txt; rm -rf $O/pmcb_$CNT; done
              python tools/pmc_bench_to_json.py $O/pmc.json $O/pmc_bench_FETCH_SIZE.txt $O/pmc_bench_WRITE_SIZE.txt | tee -a $O/summary.txt ;;
    mfmabench) # MFMA utilisation IN the pipeline: SQ_VALU_MFMA_BUSY_CYCLES (cycles, summed over the SIMDs: 32 per v_mfma_f32_32x32x16) and GRBM_GUI_ACTIVE
              # (clock cycles of the launch) in passes of their own over the one-part quick bench; tools/mfma_util.py turns them into busy fraction + effective clock
              for CNT in SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES; do
              (cd /tmp && timeout 600 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $ROOT/$O/pmcb_$CNT -o pmc -- python $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-f32-leg --no-eager-leg --no-host-fed-leg --split-batch 1 > $ROOT/$O/pmcb_$CNT.log 2>&1)
              f=$(find $O/pmcb_$CNT -name '*counter_collection.csv' | head -1); [ -n "$f" ] && python tools/pmc_summary.py $f > $O/pmc_bench_$CNT.txt; rm -rf $O/pmcb_$CNT; done
              python tools/mfma_util.py $O/pmc_bench_SQ_VALU_MFMA_BUSY_CYCLES.txt $O/pmc_bench_GRBM_GUI_ACTIVE.txt $O/kernel_stats.csv | tee $O/mfma_util.txt | tee -a $O/summary.txt ;;
    shapes) for cfg in "4 1024" "4 256" "1 512" "1 2048"; do set -- $cfg
              LAMA_BENCH_BATCH=$1 LAMA_BENCH_RES=$2 timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-f32-leg --no-eager-leg 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$1 x $2:', d['value'], 'images/s', d['ms_per_step'], 'ms')" | tee -a $O/summary.txt; done ;;
    kbench) KBENCH_ROT=6 timeout 300 python tools/kbench.py f16x3 all cold > $O/kbench_cold.log 2>&1; cp gpurun_out/kbench_cold.json $O/ 2>/dev/null; tail -30 $O/kbench_cold.log | tee -a $O/summary.txt ;;
    power)  hipcc --offload-arch=gfx950 -O3 -w tools/ubench/mfma_power.hip -o /tmp/mfma_power && /tmp/mfma_power 2>&1 | tee $O/mfma_power.txt | tail -12 | tee -a $O/summary.txt ;;
    ab)     ENVN=${ARG%%=*}; VALS=$(echo ${ARG#*=} | tr ',' ' ')
            for i in 1 2; do for v in $VALS; do
              echo -n "$ENVN=$v: " | tee -a $O/ab_$ENVN.txt
              env $ENVN=$v timeout 300 python bench.py $QUICK --lib lama_amd/lib/liblama_hip_prof.so 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['value'], 'images/s', d['ms_per_step'], 'ms', 'ffc', (d.get('roofline_ffc') or {}).get('avg_us'))" | tee -a $O/ab_$ENVN.txt; done; done
            cat $O/ab_$ENVN.txt >> $O/summary.txt ;;
    ablib)  for i in 1 2; do for v in base new; do
              if [ $v = base ]; then LIBARG="--lib $ARG"; else LIBARG=""; fi
              echo "$v: $(timeout 300 python bench.py $QUICK $LIBARG 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')" | tee -a $O/ablib.txt; done; done
            cat $O/ablib.txt >> $O/summary.txt ;;
    sh)     timeout 900 bash -c "$ARG" > $O/sh_$(echo "$ARG" | md5sum | cut -c1-6).log 2>&1; tail -40 $O/sh_$(echo "$ARG" | md5sum | cut -c1-6).log | tee -a $O/summary.txt ;;
    py)     set -- $ARG; timeout 900 python "$@" > $O/py_$(basename $1 .py).log 2>&1; tail -40 $O/py_$(basename $1 .py).log | tee -a $O/summary.txt ;;
    *)      echo "unknown step $STEP" | tee -a $O/summary.txt ;;
  esac
done
echo "== done" | tee -a $O/summary.txt
