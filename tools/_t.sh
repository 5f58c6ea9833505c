O=gpurun_out/r03m; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "winograd" 2>&1 | tail -2 | tee $O/t_wino.txt
KPROBE_ITERS=30 python tools/kprobe.py f16x3 convA wino wino128 2>&1 | grep us | tee $O/kprobe_wino.txt
LAMA_HIP_LIB=$PWD/lama_amd/lib/liblama_hip_prof.so python tools/wg_trace.py 64 4 2>&1 | tail -6 | tee $O/wg_trace.txt
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o w -- python $GRAFT_REPO_ROOT/tools/kprobe.py f16x3 wino > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
for db in $(find $O/prof -name '*.db' | head -1); do python tools/rocpd_summary.py $db $O/wino_kernel_stats.csv; done; rm -rf $O/prof
head -3 $O/wino_kernel_stats.csv | cut -c1-160
