mkdir -p gpurun_out/r04a
O=gpurun_out/r04a
P=$PWD/lama_amd/lib/liblama_hip_prof.so
for i in 1 2; do
  for ct in 1 3; do LAMA_HIP_LIB=$P LAMA_CT=$ct timeout 200 python tools/outer_ab.py up1 up2 up3 2>&1 | tail -3 | tee -a $O/ab_convt.txt; done
  for s2 in 1 3; do LAMA_HIP_LIB=$P LAMA_CW_S2=$s2 timeout 200 python tools/outer_ab.py down1 down2 down3 2>&1 | tail -3 | tee -a $O/ab_down.txt; done
done
timeout 200 python tools/outer_ab.py 2>&1 | tail -8 | tee $O/outer_product.txt
bash tools/session.sh r04a quick
bash tools/session.sh r04a tests
