#!/bin/bash
# round 2, session n: k-step timeline and ablations of the row-rolling local 3x3 kernel
mkdir -p gpurun_out/r02n
O=gpurun_out/r02n
export LAMA_HIP_LIB=$PWD/lama_amd/lib/liblama_hip_prof.so
python tools/wr_trace.py convA > $O/wr_trace_convA.txt 2>&1; cat $O/wr_trace_convA.txt
for a in 0 1 2 4 16 32 8 7 128 132 135 0; do echo -n "ROLL ABL=$a " >> $O/abl_roll.txt; LAMA_CW_ABLATE=$a KPROBE_ITERS=30 python tools/kprobe.py f16x3 convA 2>&1 | grep convA >> $O/abl_roll.txt; done
cat $O/abl_roll.txt
