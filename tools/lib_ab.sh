#!/bin/bash
# Same-box A/B of two builds of the library on tools/shape_probe.py: lib_ab.sh <old.so> <new.so> <shape> [<shape> ...]; two alternating rounds.
OLD=$1; NEW=$2; shift 2
for R in 1 2; do
  for L in $OLD $NEW; do
    echo "== round $R $(basename $L)"
    LAMA_TOOL_LIB=$L PROBE_KERNELS=1 PROBE_STEPS=10 python tools/shape_probe.py "$@" 2>&1 | grep -E "ms/step|fourier_unit"
  done
done
