mkdir -p gpurun_out/r02af
for i in 1 2; do
  echo -n "base: " >> gpurun_out/r02af/ab_head.txt; LAMA_HIP_LIB=$PWD/abtmp/base.so KPROBE_ITERS=30 python tools/kprobe.py f16x3 head 2>/dev/null | tr '\n' ' ' >> gpurun_out/r02af/ab_head.txt; echo >> gpurun_out/r02af/ab_head.txt
  echo -n "new:  " >> gpurun_out/r02af/ab_head.txt; KPROBE_ITERS=30 python tools/kprobe.py f16x3 head 2>/dev/null | tr '\n' ' ' >> gpurun_out/r02af/ab_head.txt; echo >> gpurun_out/r02af/ab_head.txt
done
cat gpurun_out/r02af/ab_head.txt
export TMPDIR=/tmp
bash tools/pmc_session.sh r02af/pmc_head "f16x3 head" "FETCH_SIZE" > gpurun_out/r02af/pmc_head.log 2>&1; grep head7 gpurun_out/r02af/pmc_head.log
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "k7 or head" 2>&1 | tail -2
