O=gpurun_out/r04d; mkdir -p $O
for v in kg1ar3 k1a3_abl1 k1a3_abl8 k1a3_abl16 k1a3_abl17 k1a3_abl20 k1a3_abl24 k1a3_abl21 k1a3_abl25 k1a3_abl28; do ./abtmp/ct_bench_$v up3 0 2>&1 | tee -a $O/ct_abl2.txt; done
