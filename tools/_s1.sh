set -x
O=gpurun_out/r03a; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "conv2d and (c384 or c192)" 2>&1 | tail -3 | tee $O/t1.txt
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "fourier_unit" 2>&1 | tail -3 | tee -a $O/t1.txt
export LAMA_HIP_LIB=$PWD/lama_amd/lib/liblama_hip_prof.so
for i in 1 2; do for v in 0 1; do echo "LAMA_GEMM_WL=$v: $(LAMA_GEMM_WL=$v KPROBE_ITERS=30 python tools/kprobe.py f16x3 conv1 fuconv fu 2>/dev/null | tr '\n' ' ')" | tee -a $O/kprobe_ab.txt; done; done
unset LAMA_HIP_LIB
bash tools/session.sh r03a "ab:LAMA_GEMM_WL=0,1" "ab:LAMA_OVERLAP_STREAMS=1,0"
timeout 900 python -m pytest tests/test_refinement_gpu.py -m gpu -x -q -k "golden" -s 2>&1 | tail -25 | tee $O/t_refine.txt
