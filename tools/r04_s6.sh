O=gpurun_out/r04f; mkdir -p $O
for i in 1 2; do for v in old ld ld_d4; do ./abtmp/wk_bench_$v 2112 0 2>&1 | tee -a $O/wk_loader.txt; done; done
