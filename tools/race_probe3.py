#!/usr/bin/env python3
"""Narrow the overlap race: the spectral branch (conv1 -> rfft2) on the side stream with selectable concurrent main-stream work."""
import sys
import torch
sys.path.insert(0, '.')
from lama_amd import _lib as L
lib = L.get_lib()
P = L.PREC_F16X3
B, H, W = 4, 32, 32
g = torch.Generator().manual_seed(0)
xg = torch.randn(B, 384, H, W, generator=g).cuda()
x1 = torch.zeros(B, 192, H, W, device='cuda')
spec = torch.zeros(B, 384, H, W // 2 + 1, device='cuda')
w1 = lib.pack_conv_weight((torch.randn(192, 384, 1, 1, generator=g) * 0.05).cuda(), None, precision=P)
b1 = torch.randn(192, generator=g).cuda()
xa = torch.randn(B, 512, H, W, generator=g).cuda(); ya = torch.empty(B, 128, H, W, device='cuda')
wa = lib.pack_conv_weight(torch.randn(128, 512, 3, 3, generator=g).cuda(), None, precision=P)
main = torch.cuda.current_stream(); side = torch.cuda.Stream()

def branch(st):
    lib.conv2d(L.view(xg), w1, L.view(x1), B, 1, bias=b1, act=L.ACT_RELU, precision=P, stream=st)
    lib.rfft2(L.view(x1), L.view(spec), B, None, st)

branch(main.cuda_stream); torch.cuda.synchronize()
ref_x1, ref_s = x1.clone(), spec.clone()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
for mode in ('side_only', 'side+convA', 'main_serial+convA'):
    bad = 0
    for it in range(N):
        x1.zero_(); spec.zero_()
        if mode == 'main_serial+convA':
            lib.conv2d(L.view(xa), wa, L.view(ya), B, 3, 1, 1, L.PAD_REFLECT, False, None, L.ACT_RELU, precision=P, stream=main.cuda_stream)
            branch(main.cuda_stream)
        else:
            side.wait_stream(main)
            branch(side.cuda_stream)
            if mode == 'side+convA':
                lib.conv2d(L.view(xa), wa, L.view(ya), B, 3, 1, 1, L.PAD_REFLECT, False, None, L.ACT_RELU, precision=P, stream=main.cuda_stream)
            main.wait_stream(side)
        torch.cuda.synchronize()
        ex, es = torch.equal(x1, ref_x1), torch.equal(spec, ref_s)
        if not (ex and es):
            bad += 1
            if bad <= 3:
                d = (spec - ref_s).abs(); idx = (d > 0).nonzero()
                print(mode, it, 'x1 ok' if ex else 'x1 BAD', 'spec n', idx.shape[0], idx[0].tolist() if idx.shape[0] else None, idx[-1].tolist() if idx.shape[0] else None, flush=True)
    print(mode, 'mismatching iterations:', bad, 'of', N, flush=True)
