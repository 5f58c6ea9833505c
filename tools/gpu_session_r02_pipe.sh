#!/bin/bash
# round 2: the local convs as a chain of their own on the second stream (ffc.SidePipe) -- tests, stress, A/B, timeline
O=gpurun_out/r02pipe
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "overlap_streams or cooperative" 2>&1 | tail -4 | tee $O/summary.txt
timeout 600 python tools/pipeline_stress.py 400 2>&1 | tail -3 | tee -a $O/summary.txt
timeout 900 python -m pytest tests/test_generator_gpu.py -m gpu -x -q 2>&1 | tail -4 | tee -a $O/summary.txt
for f in 1 0 1 0; do echo -n "LAMA_PIPELINE_LOCAL=$f " >> $O/ab_pipe.txt; LAMA_PIPELINE_LOCAL=$f python bench.py --no-f32-leg --no-cpu-baseline --no-eager-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" >> $O/ab_pipe.txt; done
cat $O/ab_pipe.txt
(cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$O/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-f32-leg --no-eager-leg > $GRAFT_REPO_ROOT/$O/bench.log 2>&1)
for db in $(find $O/prof -name '*.db' | head -1); do python tools/timeline.py $db $O/timeline.txt 4; done
rm -rf $O/prof
sed -n 40,62p $O/timeline.txt; tail -1 $O/timeline.txt
