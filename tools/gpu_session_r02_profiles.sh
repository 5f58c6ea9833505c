#!/bin/bash
# round 2 evidence session: rocprofv3 --kernel-trace --stats of the default bench, PMC passes (FETCH_SIZE / WRITE_SIZE / MFMA busy, each in
# its own run) over tools/kprobe.py at BASELINE configs[1] layer shapes, MFMA power micro-benchmark.  usage: bash tools/gpu_session_r02_profiles.sh
TAG=r02prof
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 -w tools/ubench/mfma_power.hip -o /tmp/mfma_power && /tmp/mfma_power > $OUT/mfma_power.txt 2>&1
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-f32-leg --no-eager-leg > $GRAFT_REPO_ROOT/$OUT/prof_bench.log 2>&1)
for db in $(find $OUT/prof -name '*.db' | head -1); do python tools/rocpd_summary.py $db $OUT/kernel_stats.csv; done
find $OUT/prof -name '*stats*.csv' | head -3
for f in $(find $OUT/prof -name '*kernel_stats.csv' | head -1); do cp $f $OUT/rocprofv3_kernel_stats_raw.csv; done
head -16 $OUT/kernel_stats.csv | cut -c1-160
rm -rf $OUT/prof
bash tools/pmc_session.sh $TAG/pmc_all "f16x3 convA convB convBf rfft irfft down1 down2 down3 up1 up2 up3 stem head" "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT" > $OUT/pmc_all.log 2>&1
bash tools/pmc_session.sh $TAG/pmc_conv1 "f16x3 conv1" "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" > $OUT/pmc_conv1.log 2>&1
bash tools/pmc_session.sh $TAG/pmc_fuconv "f16x3 fuconv" "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" > $OUT/pmc_fuconv.log 2>&1
tail -30 $OUT/pmc_all.log | cut -c1-200
