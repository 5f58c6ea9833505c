#!/bin/bash
# round 2: s_setprio 3 in the spectral branch's kernels (rfft2 / spectral GEMM / irfft2 share SIMDs with the local conv) -- A/B + timeline
O=gpurun_out/r02prio
mkdir -p $O
export TMPDIR=/tmp
export LAMA_HIP_LIB=$PWD/lama_amd/lib/liblama_hip_prof.so
for f in 1 0 1 0; do echo -n "LAMA_SIDE_PRIO=$f " >> $O/ab_prio.txt; LAMA_SIDE_PRIO=$f python bench.py --no-f32-leg --no-cpu-baseline --no-eager-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" >> $O/ab_prio.txt; done
cat $O/ab_prio.txt
unset LAMA_HIP_LIB
(cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$O/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-f32-leg --no-eager-leg > $GRAFT_REPO_ROOT/$O/bench.log 2>&1)
for db in $(find $O/prof -name '*.db' | head -1); do python tools/timeline.py $db $O/timeline.txt 4; done
rm -rf $O/prof
sed -n 40,56p $O/timeline.txt; tail -1 $O/timeline.txt
