#!/usr/bin/env python3
"""K-sweep of a 1x1 conv: time vs input channels (slope = per-stage cost, intercept = fixed per-launch / per-workgroup cost)."""
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
for (cout, H, W, k) in ((384, 64, 33, 1), (192, 64, 64, 1), (128, 64, 64, 3)):
    for cin in (32, 64, 128, 256, 512, 1024):
        x = torch.randn(B, cin, H, W, generator=g).cuda()
        w = torch.randn(cout, cin, k, k, generator=g).cuda()
        wp = lib.pack_conv_weight(w, None, precision=prec)
        y = torch.empty(B, cout, H, W, device='cuda')
        fn = lambda: lib.conv2d(L.view(x), wp, L.view(y), B, k, 1, k // 2, L.PAD_REFLECT, False, None, L.ACT_RELU, precision=prec, stream=st)
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            fn()
        b.record()
        torch.cuda.synchronize()
        print(f'k{k} cout={cout} {H}x{W} cin={cin:5d}: {a.elapsed_time(b) * 1e3 / 20:8.2f} us', flush=True)
