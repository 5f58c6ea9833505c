#!/usr/bin/env python3
"""K-sweeps of the high-resolution layers: stride-2 3x3 (down1 shape), transposed 3x3 (up3 shape), stem/head 7x7."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lama_amd import _lib as L  # noqa: E402

lib = L.get_lib()
prec = L.PREC_NAMES[sys.argv[1] if len(sys.argv) > 1 else 'f16x3']
st = torch.cuda.current_stream().cuda_stream
g = torch.Generator().manual_seed(0)
B = 8


def run(name, cin, cout, k, H, W, stride=1, tr=False):
    x = torch.randn(B, cin, H, W, generator=g).cuda()
    w = (torch.randn(cin, cout, k, k, generator=g) if tr else torch.randn(cout, cin, k, k, generator=g)).cuda()
    s2 = 2 if tr else stride
    wp = lib.pack_conv_weight(w, None, stride=s2, transposed=tr, precision=prec)
    Ho, Wo = (2 * H, 2 * W) if tr else ((H + 2 * (k // 2) - k) // s2 + 1, (W + 2 * (k // 2) - k) // s2 + 1)
    y = torch.empty(B, cout, Ho, Wo, device='cuda')
    fn = lambda: lib.conv2d(L.view(x), wp, L.view(y), B, k, s2, 1 if tr else k // 2, L.PAD_ZERO if tr else L.PAD_REFLECT, tr, None, L.ACT_RELU,
                            precision=prec, stream=st)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        fn()
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1e2
    mb = 4e-6 * B * (cin * H * W + cout * Ho * Wo)
    print(f'{name:8s} cin={cin:4d} cout={cout:4d} {H}x{W}: {us:8.1f} us   hbm {mb:7.1f} MB -> {mb / us * 1e3 / 1e3:5.2f} TB/s', flush=True)


for cin in (16, 32, 64, 128):
    run('down s2', cin, 128, 3, 512, 512, stride=2)
for cin in (16, 64, 128):
    run('up T', cin, 64, 3, 256, 256, tr=True)
for cin in (16, 64):
    run('conv3x3', cin, 64, 3, 512, 512)
run('stem', 4, 64, 7, 512, 512)
for cin in (16, 64):
    run('head', cin, 3, 7, 512, 512)
