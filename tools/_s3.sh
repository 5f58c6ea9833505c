O=gpurun_out/r03c; mkdir -p $O
nproc | tee $O/nproc.txt
timeout 1200 python tools/refine_diag.py 1024 4 f32 2>&1 | grep -v -i warn | tail -20 | tee $O/refine_diag_f32.txt
