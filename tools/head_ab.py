#!/usr/bin/env python3
"""Same-box A/B of the head kernel (7x7 64 -> 3 + sigmoid at 8 x 512^2) between two builds of the library: the in-tree one and LAMA_AB_BASE.
Rotated operands (the 537 MB input does not sit in the Infinity Cache anyway), HIP events around every launch, medians; also checks that the two
outputs agree.  usage: LAMA_AB_BASE=abtmp/base_liblama_hip.so python tools/head_ab.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lama_amd import _lib as L
from lama_amd import build as B
dev = 'cuda'
g = torch.Generator().manual_seed(0)
x = [torch.rand(8, 64, 512, 512, generator=g).to(dev) for _ in range(2)]
w = (torch.randn(3, 64, 7, 7, generator=g) * 0.02).to(dev)
bias = torch.randn(3, generator=g).to(dev)
st = torch.cuda.current_stream().cuda_stream
outs = {}
libs = [('new', B.LIB), ('base', os.environ.get('LAMA_AB_BASE', B.LIB)), ('new', B.LIB), ('base', os.environ.get('LAMA_AB_BASE', B.LIB))]
for name, path in libs:
    lib = L.LamaLib(path)
    for prec in ('f16x3', 'bf16x3'):
        P = L.PREC_NAMES[prec]
        wp = lib.pack_conv_weight(w, None, precision=P)
        y = torch.empty(8, 3, 512, 512, device=dev)
        ts = []
        for i in range(14):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            lib.conv2d(L.view(x[i & 1]), wp, L.view(y), 8, 7, 1, 3, L.PAD_REFLECT, False, bias, L.ACT_SIGMOID, precision=P, stream=st)
            b.record()
            torch.cuda.synchronize()
            if i >= 4:
                ts.append(a.elapsed_time(b) * 1e3)
        ts.sort()
        key = (prec,)
        if key in outs:
            d = float((outs[key] - y).abs().max())
        else:
            outs[key] = y.clone(); d = 0.0
        print(f'{name:5s} {prec:7s} head7x7 8x64x512x512: median {ts[len(ts)//2]:.1f} us  min {ts[0]:.1f}  max-abs diff to the first build {d:.2e}', flush=True)
