O=gpurun_out/r03b; mkdir -p $O
export TMPDIR=/tmp
export LAMA_HIP_LIB=$PWD/lama_amd/lib/liblama_hip_prof.so
for i in 1 2; do for v in 0 1; do echo "LAMA_GEMM_WL=$v: $(LAMA_GEMM_WL=$v KPROBE_ITERS=30 python tools/kprobe.py f16x3 conv1 fuconv fu 2>/dev/null | tr '\n' ' ')" | tee -a $O/kprobe_ab.txt; done; done
python tools/gl_trace.py fuconv 6 2>&1 | tail -12 | tee $O/gl_trace_fuconv.txt
python tools/gl_trace.py conv1 6 2>&1 | tail -12 | tee $O/gl_trace_conv1.txt
unset LAMA_HIP_LIB
timeout 900 python -m pytest tests/test_refinement_gpu.py -m gpu -q -k "golden and 1024" -s 2>&1 | grep -E "scale|passed|failed|Error" | tee $O/t_refine.txt
