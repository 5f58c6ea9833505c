#!/bin/bash
# A/B two builds of the library on the same box: abtmp/base.so vs the in-tree build.  usage: tools/ab.sh [kprobe names...]
for i in 1 2; do for v in base new; do
  if [ $v = base ]; then export LAMA_HIP_LIB=$PWD/abtmp/base.so; else unset LAMA_HIP_LIB; fi
  echo "== $v: $(python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-f32-leg --no-eager-leg 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')"
  [ -n "$1" ] && [ $i = 1 ] && python tools/kprobe.py f16x3 "$@" 2>/dev/null | tr '\n' ' ' && echo
done; done
