#!/bin/bash
# Recompile ONE translation unit (default conv_f16x3.hip) into both libraries and relink: ~1.5 min instead of a full build.  The digest
# stamps are refreshed so that lama_amd.build does not rebuild on import.  usage: tools/quick_build.sh [source.hip ...]
cd "$(dirname "$0")/.."
SRCS=${@:-conv_f16x3.hip}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Iinclude -Ilama_amd/csrc -Xclang -target-feature -Xclang -packed-fp32-ops"
for s in $SRCS; do
  hipcc $FLAGS -c lama_amd/csrc/$s -o lama_amd/lib/obj/liblama_hip/$s.o &
  hipcc $FLAGS -DLAMA_PROFILING -c lama_amd/csrc/$s -o lama_amd/lib/obj/liblama_hip_prof/$s.o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o lama_amd/lib/liblama_hip.so lama_amd/lib/obj/liblama_hip/*.o
hipcc --offload-arch=gfx950 -shared -fPIC -o lama_amd/lib/liblama_hip_prof.so lama_amd/lib/obj/liblama_hip_prof/*.o
python - <<'PY'
from lama_amd import build as B
import os
srcs = [os.path.join(B.CSRC, s) for s in B.SOURCES]
open(os.path.join(B.LIBDIR, 'liblama_hip.sha256'), 'w').write(B._digest(srcs + B.HEADERS, []))
open(os.path.join(B.LIBDIR, 'liblama_hip_prof.sha256'), 'w').write(B._digest(srcs + [os.path.join(B.CSRC, s) for s in B.PROF_ONLY_SOURCES] + B.HEADERS, ['-DLAMA_PROFILING']))
PY
echo rebuilt
