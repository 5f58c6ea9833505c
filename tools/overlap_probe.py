#!/usr/bin/env python3
"""Does a pure streaming kernel (add_kernel: the traffic pattern of wino_out_kernel) fill the memory-idle LDS phase of rfft2_ip64_kernel when both are
in flight at once?  Pairs on two streams against the same launches back to back on one stream (8 x 192 planes of 64 x 64; rotated operand sets).
usage: overlap_probe.py [n_iter]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _toollib  # noqa: E402,F401  (LAMA_TOOL_LIB=<path>: another build of the library)
from lama_amd import _lib as L  # noqa: E402
from lama_amd import ffc as F  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device('cuda')
lib = F._DEFAULT_EXEC.lib
NR = 6
xs = [torch.randn(8, 192, 64, 64, device=dev) for _ in range(NR)]
specs = [torch.empty(8, 384, 64, 33, device=dev) for _ in range(NR)]
a = [torch.randn(8, 256, 64, 64, device=dev) for _ in range(NR)]       # add: 3 x 33.5 MB = wino_out's 67 MB + its output
b = [torch.randn(8, 256, 64, 64, device=dev) for _ in range(NR)]
o = [torch.empty(8, 256, 64, 64, device=dev) for _ in range(NR)]
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def run(mode):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        k = i % NR
        if mode in ('fft', 'serial', 'pair'):
            lib.rfft2(L.view(xs[k]), L.view(specs[k]), 8, None, s1.cuda_stream)
        if mode in ('add', 'serial'):
            lib.add(L.view(a[k]), L.view(b[k]), L.view(o[k]), 8, s1.cuda_stream)
        if mode == 'pair':
            lib.add(L.view(a[k]), L.view(b[k]), L.view(o[k]), 8, s2.cuda_stream)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for mode in ('fft', 'add', 'serial', 'pair', 'fft', 'add', 'serial', 'pair'):
    run(mode)
    print(f'{mode:7s} {run(mode):7.2f} us per iteration', flush=True)
