O=gpurun_out/r04c; mkdir -p $O
for n in up3 up2 up1; do
  for v in kg1ar6 kg1ar3 kg2ar6 kg2ar3; do ./abtmp/ct_bench_$v $n 0 2>&1 | tee -a $O/ct_var.txt; done
done
