#!/bin/bash
# scheduling variants of the weights-in-registers 3x3 kernel (LAMA_CW_SCHED: 0 compiler order, 1 phases, 2 MFMA-shadow interleave)
for s in 2 1 0 2; do
  echo -n "SCHED=$s  "; LAMA_CW_SCHED=$s KPROBE_ITERS=30 python tools/kprobe.py f16x3 convA 2>&1 | grep convA
done
for wr in 1 0; do
  echo -n "WR=$wr  "; LAMA_CONV_WR=$wr KPROBE_ITERS=30 python tools/kprobe.py f16x3 convA convB conv1 fuconv 2>&1 | grep " us" | tr '\n' ' '; echo
done
ABLS="0 1 2 4 7 8" bash tools/wr_abl.sh
