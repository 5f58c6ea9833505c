#!/bin/bash
# timing ablations of the bottleneck 3x3 conv (LAMA_CB_ABLATE bits: 1 no MFMA, 2 no fragment reads, 4 no staging, 8 no barrier)
for a in 0 4 16 32 6 15; do
  echo -n "ABL=$a  "; LAMA_CB_ABLATE=$a KPROBE_ITERS=20 python tools/kprobe.py bf16x3 convA 2>&1 | grep convA
done
