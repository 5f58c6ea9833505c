"""Build the host-emulated copy of the kernels (tests/hipemu/build/liblama_emu.so).  TEST-ONLY:
the same lama_amd/csrc/*.hip sources, compiled for x86 against tests/hipemu/hip/hip_runtime.h."""
import hashlib
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, 'lama_amd', 'csrc')
OUT = os.path.join(HERE, 'build')
LIB = os.path.join(OUT, 'liblama_emu.so')


def _cxx():
    for c in ('/opt/rocm/lib/llvm/bin/clang++', '/usr/bin/clang++'):
        if os.path.exists(c):
            return c
    raise RuntimeError('clang++ (ext_vector_type support) not found')


def build(force=False):
    import fcntl
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, '.build.lock'), 'w') as lock:      # pytest-xdist workers get here at the same time
        fcntl.flock(lock, fcntl.LOCK_EX)
        return _build(force)


def _build(force):
    srcs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hip') and f != 'debug_probes.hip')
    deps = srcs + [os.path.join(CSRC, 'common.h'), os.path.join(CSRC, 'conv_split3.inc'), os.path.join(CSRC, 'conv_wreg_dev.inc'), os.path.join(CSRC, 'conv_ws_dev.inc'), os.path.join(CSRC, 'gemm_wk_dev.inc'), os.path.join(CSRC, 'conv_stem_dev.inc'), os.path.join(CSRC, 'conv_head_dev.inc'), os.path.join(CSRC, 'wino_dev.inc'), os.path.join(CSRC, 'convt_dev.inc'), os.path.join(CSRC, 'wino_out_dev.inc'), os.path.join(CSRC, 'fft_mr_dev.inc'),
                   os.path.join(CSRC, 'conv_wreg_host.inc'), os.path.join(ROOT, 'include', 'lama_hip.h'),
                   os.path.join(HERE, 'hip', 'hip_runtime.h'), os.path.join(HERE, 'hipemu_runtime.cpp')]
    h = hashlib.sha256()
    for d in deps:
        h.update(open(d, 'rb').read())
    stamp = os.path.join(OUT, 'stamp')
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == h.hexdigest():
        return LIB
    cxx = _cxx()
    objs = []
    procs = []
    for s in srcs + [os.path.join(HERE, 'hipemu_runtime.cpp')]:
        o = os.path.join(OUT, os.path.basename(s) + '.o')
        cmd = [cxx, '-x', 'c++', '-std=c++17', '-O2', '-fPIC', '-DLAMA_PROFILING', '-Wno-unknown-attributes', '-Wno-ignored-attributes',
               '-I' + HERE, '-I' + os.path.join(ROOT, 'include'), '-I' + CSRC, '-c', s, '-o', o]
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(o)
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode:
            raise RuntimeError(f'emu compile failed for {s}:\n{out}')
    r = subprocess.run([cxx, '-shared', '-fPIC', '-o', LIB, *objs], capture_output=True, text=True)
    if r.returncode:
        raise RuntimeError(r.stdout + r.stderr)
    open(stamp, 'w').write(h.hexdigest())
    return LIB


if __name__ == '__main__':
    print(build(force=True))
