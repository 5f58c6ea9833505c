#!/bin/bash
# timing ablations of the weights-in-registers 3x3 kernel (LAMA_CW_ABLATE bits: 1 no A loads, 2 no B reads, 4 no staging, 8 no MFMA)
for a in ${ABLS:-0 1 2 4 3 7 8 12 6}; do
  echo -n "ABL=$a  "; LAMA_CW_ABLATE=$a KPROBE_ITERS=30 python tools/kprobe.py f16x3 convA 2>&1 | grep convA
done
