O=gpurun_out/r04e; mkdir -p $O
for n in up3 up1; do for v in kg1ar3 k1a3_st500 k1a3_st1000 k1a3_st2000 k1a3_st3000; do ./abtmp/ct_bench_$v $n 0 2>&1 | tee -a $O/ct_stagger.txt; done; done
