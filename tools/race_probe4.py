#!/usr/bin/env python3
"""Full bottleneck FFC layer with the side stream, step by step, snapshotting the spectra right after each side kernel."""
import sys
import torch
sys.path.insert(0, '.')
import torch.nn as nn
from lama_amd import ffc as F, _lib as L
lib = L.get_lib()
torch.manual_seed(0)
B, H, W = 4, 32, 32
wf = W // 2 + 1
lay = F.FFC_BN_ACT(512, 512, kernel_size=3, ratio_gin=0.75, ratio_gout=0.75, padding=1, norm_layer=nn.BatchNorm2d,
                   activation_layer=nn.ReLU, enable_lfu=False).cuda()
lay.train(False)
pk = lay._pack(); st_ = lay.ffc.convg2g; sp = st_._packed; fuw, fub = st_.fu._pack()
P = lay.precision
src = torch.randn(B, 512, H, W, device='cuda'); resid = torch.randn(B, 512, H, W, device='cuda')
dst = torch.zeros_like(src)
x1 = torch.zeros(B, 192, H, W, device='cuda'); t = torch.zeros_like(x1)
s1 = torch.zeros(B, 384, H, wf, device='cuda'); s2 = torch.zeros_like(s1)
snap1 = torch.zeros_like(s1); snap2 = torch.zeros_like(s1)
main = torch.cuda.current_stream(); side = torch.cuda.Stream()
variant = sys.argv[2] if len(sys.argv) > 2 else 'full'

def run(use_side):
    ss = side if use_side else main
    if use_side: side.wait_stream(main)
    s = ss.cuda_stream
    lib.conv2d(L.view(src, 128, 384), sp['w1'], L.view(x1), B, 1, bias=sp['b1'], act=L.ACT_RELU, precision=P, stream=s)
    lib.rfft2(L.view(x1), L.view(s1), B, None, s)
    with torch.cuda.stream(ss): snap1.copy_(s1)
    if variant != 'nofu':
        lib.conv2d(L.view(s1), fuw, L.view(s2), B, 1, bias=fub, act=L.ACT_RELU, precision=P, stream=s)
        with torch.cuda.stream(ss): snap2.copy_(s2)
        lib.irfft2(L.view(s2), L.view(x1), L.view(t), B, None, s)
    lib.conv2d(L.view(src), pk['w_lout'], L.view(dst, 0, 128), B, 3, 1, 1, L.PAD_REFLECT, False, pk['b_l'], L.ACT_RELU, L.view(resid, 0, 128), precision=P, stream=main.cuda_stream)
    if use_side: main.wait_stream(side)
    if variant != 'nofu':
        lib.conv2d(L.view(src, 0, 128), pk['w_l2g'], L.view(dst, 128, 384), B, 3, 1, 1, L.PAD_REFLECT, False, pk['b_g'], L.ACT_RELU, L.view(resid, 128, 384),
                   x2=L.view(t), w2_packed=sp['w2'], precision=P, stream=main.cuda_stream)

run(False); torch.cuda.synchronize()
ref = dict(s1=s1.clone(), s2=s2.clone(), t=t.clone(), x1=x1.clone(), dst=dst.clone())
bad = 0
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
for it in range(N):
    for b_ in (dst, x1, t, s1, s2, snap1, snap2): b_.zero_()
    run(True); torch.cuda.synchronize()
    res = {k: torch.equal(v, ref[k]) for k, v in (('x1', x1), ('s1', s1), ('s2', s2), ('t', t), ('dst', dst))}
    res['snap1'] = torch.equal(snap1, ref['s1']); res['snap2'] = torch.equal(snap2, ref['s2'])
    if not all(res.values()):
        bad += 1
        if bad <= 6: print(it, {k: v for k, v in res.items() if not v}, flush=True)
print(variant, 'mismatching iterations:', bad, 'of', N)
