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
SOURCES = ['conv_mfma.hip', 'conv_bf16x3.hip', 'conv_f16x3.hip', 'fft.hip', 'elementwise.hip']
HEADERS = [os.path.join(CSRC, 'common.h'), os.path.join(CSRC, 'conv_split3.inc'), os.path.join(CSRC, 'conv_wreg_dev.inc'), os.path.join(CSRC, 'conv_ws_dev.inc'), os.path.join(CSRC, 'conv_stem_dev.inc'), os.path.join(CSRC, 'conv_head_dev.inc'),
           os.path.join(CSRC, 'conv_wreg_host.inc'), os.path.join(ROOT, 'include', 'lama_hip.h')]
HIPCC_FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fno-gpu-rdc',
               '-I' + os.path.join(ROOT, 'include'), '-I' + CSRC]


def _hipcc() -> str:
    for cand in (shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError('hipcc not found: liblama_hip.so cannot be built')


def _digest(paths) -> str:
    h = hashlib.sha256()
    for p in paths:
        with open(p, 'rb') as f:
            h.update(f.read())
    h.update(' '.join(HIPCC_FLAGS).encode())
    return h.hexdigest()


def _compile(args):
    src, obj, hipcc = args
    cmd = [hipcc, *HIPCC_FLAGS, '-c', src, '-o', obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f'hipcc failed for {src}:\n{r.stdout}\n{r.stderr}')
    return obj


def build(force: bool = False, verbose: bool = True) -> str:
    """Compile every HIP source for gfx950 and link lama_amd/lib/liblama_hip.so; returns its path."""
    os.makedirs(LIBDIR, exist_ok=True)
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    stamp = os.path.join(LIBDIR, 'liblama_hip.sha256')
    dig = _digest(srcs + HEADERS)
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read().strip() == dig:
        return LIB
    hipcc = _hipcc()
    objdir = os.path.join(LIBDIR, 'obj')
    os.makedirs(objdir, exist_ok=True)
    jobs = [(s, os.path.join(objdir, os.path.basename(s) + '.o'), hipcc) for s in srcs]
    if verbose:
        print(f'[lama_amd.build] hipcc {len(jobs)} sources for gfx950 ...', file=sys.stderr)
    with concurrent.futures.ThreadPoolExecutor(max_workers=len(jobs)) as ex:
        objs = list(ex.map(_compile, jobs))
    r = subprocess.run([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB, *objs], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f'link failed:\n{r.stdout}\n{r.stderr}')
    with open(stamp, 'w') as f:
        f.write(dig)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
