"""Build liblama_hip.so (gfx950) in-tree with hipcc.  Usage: ``python -m lama_amd.build [--force]``.

The shared object is written next to the sources (``lama_amd/lib/liblama_hip.so``) so that it
travels with a snapshot of the repo; it is git-ignored, the sources are the history.
"""
from __future__ import annotations

import concurrent.futures
import hashlib
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, 'csrc')
LIBDIR = os.path.join(PKG, 'lib')
LIB = os.path.join(LIBDIR, 'liblama_hip.so')
# Same sources with -DLAMA_PROFILING: kernel-selection overrides, timing ablations and timeline tracers read from the environment.
# Only tools/ and the forced-path GPU tests load it (tools/_toollib.py, bench.py --lib, LamaLib(path)); the product never does.
LIB_PROF = os.path.join(LIBDIR, 'liblama_hip_prof.so')
SOURCES = ['conv_mfma.hip', 'conv_bf16x3.hip', 'conv_f16x3.hip', 'conv_f16.hip', 'fft.hip', 'elementwise.hip', 'refine.hip', 'metrics.hip']
# Every translation unit is compiled WITHOUT packed-fp32 VALU instructions (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32).
# Measured on MI355X / ROCm 7.2 (the probes -- tools/race_probe1-9.py, overlap_stress.py -- are in the git history): a v_pk_*_f32 with an op_sel half-swizzle returns wrong
# results while a wave of ANOTHER kernel executes MFMA instructions on the same SIMD (an inline-asm probe of that one instruction
# fails 60 / 60 next to a bare MFMA loop, the plain forms pass).  hipcc's SLP vectoriser emits thousands of them for the float2
# butterflies of the FFT kernels, which therefore produced wrong planes in up to 99 % of the runs next to a convolution on a second
# stream -- and 0 of 100 000 without them.  Costs nothing measurable (bench: 555 vs 559 images/s, inside the run-to-run noise).
NO_PACKED_FP32 = ['-Xclang', '-target-feature', '-Xclang', '-packed-fp32-ops']
PROF_ONLY_SOURCES = ['debug_probes.hip']      # lama_debug_mfma_peak (bench.py peak_sustained) + the opcode probes of the co-residency investigation: built WITH packed fp32, on purpose
PER_SOURCE_FLAGS = {}                        # filled below: NO_PACKED_FP32 for every product source
HIPCC_FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fno-gpu-rdc',
               '-I' + os.path.join(ROOT, 'include'), '-I' + CSRC]


def _hipcc() -> str:
    for cand in (shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError('hipcc not found: liblama_hip.so cannot be built')


def _digest(paths, extra_flags=()) -> str:
    h = hashlib.sha256()
    for p in paths:
        with open(p, 'rb') as f:
            h.update(f.read())
    h.update(' '.join(list(HIPCC_FLAGS) + list(extra_flags) + [repr(sorted(PER_SOURCE_FLAGS.items()))]).encode())
    return h.hexdigest()


def _compile(args):
    src, obj, hipcc, extra = args
    cmd = [hipcc, *HIPCC_FLAGS, *extra, *PER_SOURCE_FLAGS.get(os.path.basename(src), []), '-c', src, '-o', obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f'hipcc failed for {src}:\n{r.stdout}\n{r.stderr}')
    return obj


for _s in SOURCES:
    PER_SOURCE_FLAGS[_s] = NO_PACKED_FP32


def _include_closure(path, seen=None):
    """The file and every `#include "..."` it reaches under csrc/ or include/ (what an object file really depends on)."""
    import re
    seen = set() if seen is None else seen
    if path in seen:
        return seen
    seen.add(path)
    for name in re.findall(r'^\s*#\s*include\s+"([^"]+)"', open(path).read(), re.M):
        for base in (os.path.dirname(path), CSRC, os.path.join(ROOT, 'include')):
            cand = os.path.join(base, name)
            if os.path.exists(cand):
                _include_closure(cand, seen)
                break
    return seen


def _build_one(lib: str, extra_flags, force: bool, verbose: bool, extra_sources=()) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    srcs = [os.path.join(CSRC, s) for s in list(SOURCES) + list(extra_sources)]
    tag = os.path.splitext(os.path.basename(lib))[0]
    stamp = os.path.join(LIBDIR, tag + '.sha256')
    dig = _digest(sorted(set().union(*[_include_closure(s) for s in srcs])), extra_flags)   # every file any source includes: no list to maintain
    if not force and os.path.exists(lib) and os.path.exists(stamp) and open(stamp).read().strip() == dig:
        return lib
    hipcc = _hipcc()
    objdir = os.path.join(LIBDIR, 'obj', tag)
    os.makedirs(objdir, exist_ok=True)
    # per-object stamps over the include closure of each source: an edit recompiles only the translation units that see it
    jobs, objs, stamps = [], [], {}
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s) + '.o')
        odig = _digest(sorted(_include_closure(s)), extra_flags)
        objs.append(o)
        if force or not (os.path.exists(o) and os.path.exists(o + '.sha256') and open(o + '.sha256').read().strip() == odig):
            jobs.append((s, o, hipcc, extra_flags))
            stamps[o] = odig
    if verbose:
        print(f'[lama_amd.build] hipcc {len(jobs)} of {len(srcs)} sources for gfx950 -> {os.path.basename(lib)} ...', file=sys.stderr)
    if jobs:
        with concurrent.futures.ThreadPoolExecutor(max_workers=len(jobs)) as ex:
            list(ex.map(_compile, jobs))
    for o, d in stamps.items():
        with open(o + '.sha256', 'w') as f:
            f.write(d)
    r = subprocess.run([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib, *objs], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f'link failed:\n{r.stdout}\n{r.stderr}')
    with open(stamp, 'w') as f:
        f.write(dig)
    return lib


def build(force: bool = False, verbose: bool = True, profiling: bool = True) -> str:
    """Compile every HIP source for gfx950 and link lama_amd/lib/liblama_hip.so (the product) and, with ``profiling``,
    lama_amd/lib/liblama_hip_prof.so (same sources + -DLAMA_PROFILING, for tools/ and the forced-path tests); returns the
    product's path."""
    import fcntl
    os.makedirs(LIBDIR, exist_ok=True)
    with open(os.path.join(LIBDIR, '.build.lock'), 'w') as lock:      # several processes may get here at once (pytest -n, torchrun ranks)
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not profiling:
            return _build_one(LIB, [], force, verbose)
        with concurrent.futures.ThreadPoolExecutor(max_workers=2) as ex:
            f1 = ex.submit(_build_one, LIB, [], force, verbose)
            f2 = ex.submit(_build_one, LIB_PROF, ['-DLAMA_PROFILING'], force, verbose, PROF_ONLY_SOURCES)
            f2.result()
            return f1.result()


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
