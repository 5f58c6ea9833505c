O=gpurun_out/r04b; mkdir -p $O
for n in up3 up1; do
  for v in base abl1 abl2 abl4 abl8 abl12 abl16 abl32 abl64 abl29; do ./abtmp/ct_bench_$v $n 0 2>&1 | tee -a $O/ct_abl.txt; done
  ./abtmp/ct_bench_base $n 1 | tee -a $O/ct_abl.txt
  ./abtmp/ct_bench_base $n 0 256 | tee -a $O/ct_abl.txt
done
./abtmp/ct_bench_base up2 0 | tee -a $O/ct_abl.txt
bash tools/session.sh r04b ab:LAMA_CT=1,3
