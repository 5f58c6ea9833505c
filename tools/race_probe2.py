#!/usr/bin/env python3
"""rfft2 alone, repeated, with a big conv running concurrently on another stream: does the spectrum ever differ?"""
import sys
import torch
sys.path.insert(0, '.')
from lama_amd import _lib as L
lib = L.get_lib()
B, C, H, W = 4, int(sys.argv[1]) if len(sys.argv) > 1 else 192, 32, 32
x = torch.randn(B, C, H, W, device='cuda')
spec = torch.zeros(B, 2 * C, H, W // 2 + 1, device='cuda')
main = torch.cuda.current_stream()
side = torch.cuda.Stream()
lib.rfft2(L.view(x), L.view(spec), B, None, main.cuda_stream)
torch.cuda.synchronize()
ref = spec.clone()
# the concurrent kernel: bottleneck local conv
xa = torch.randn(B, 512, H, W, device='cuda'); ya = torch.empty(B, 128, H, W, device='cuda')
wp = lib.pack_conv_weight(torch.randn(128, 512, 3, 3, device='cuda'), None, precision=L.PREC_F16X3)
bad = 0
for mode in ('alone', 'concurrent'):
    bad = 0
    for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 1000):
        spec.zero_()
        if mode == 'concurrent':
            side.wait_stream(main)
            lib.conv2d(L.view(xa), wp, L.view(ya), B, 3, 1, 1, L.PAD_REFLECT, False, None, L.ACT_RELU, precision=L.PREC_F16X3, stream=main.cuda_stream)
            lib.rfft2(L.view(x), L.view(spec), B, None, side.cuda_stream)
            main.wait_stream(side)
        else:
            lib.rfft2(L.view(x), L.view(spec), B, None, main.cuda_stream)
        torch.cuda.synchronize()
        if not torch.equal(spec, ref):
            bad += 1
            d = (spec - ref).abs(); idx = (d > 0).nonzero()
            if bad <= 4: print(mode, it, 'n', idx.shape[0], 'first', idx[0].tolist(), 'last', idx[-1].tolist(), 'max', float(d.max()), flush=True)
    print(mode, 'mismatching iterations:', bad, flush=True)
