#!/usr/bin/env python3
"""Second round of the co-residency hunt (see race_probe5.py): what exactly breaks the conv1 -> rfft2 -> GEMM -> irfft2 chain on
the side stream when the main stream is busy?  Every variant runs the chain on the side stream; the knobs are
  busy   : what runs concurrently on the main stream: conv (the local 3x3 conv), torch (a long elementwise torch kernel), none
  sync   : none | event (record + wait an event on the side stream between consecutive kernels) | flush (a tiny torch kernel
           between consecutive kernels)
  chain  : full | fft (rfft2(x1) -> irfft2(s1): no GEMM kernels) | torchchain (pure torch ops: a = x * 2; b = a + 1; c = b * b)
"""
import sys
import torch
sys.path.insert(0, '.')
import torch.nn as nn
from lama_amd import ffc as F, _lib as L

lib = L.get_lib()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
H = W = 64
B = 8
wf = W // 2 + 1
torch.manual_seed(0)
lay = F.FFC_BN_ACT(512, 512, kernel_size=3, ratio_gin=0.75, ratio_gout=0.75, padding=1, norm_layer=nn.BatchNorm2d,
                   activation_layer=nn.ReLU, enable_lfu=False).cuda()
lay.train(False)
pk = lay._pack(); st_ = lay.ffc.convg2g; sp = st_._packed; fuw, fub = st_.fu._pack()
P = lay.precision
SENT = 12345.0
srcs = [torch.randn(B, 512, H, W, device='cuda') for _ in range(3)]
dst = torch.empty(B, 512, H, W, device='cuda')
x1 = torch.empty(B, 192, H, W, device='cuda'); t = torch.empty_like(x1)
s1 = torch.empty(B, 384, H, wf, device='cuda'); s2 = torch.empty_like(s1)
big = torch.randn(64 * 1024 * 1024, device='cuda'); big2 = torch.empty_like(big)
ta = torch.empty_like(x1); tb = torch.empty_like(x1); tc = torch.empty_like(x1)
main = torch.cuda.current_stream(); side = torch.cuda.Stream()
tiny = torch.zeros(64, device='cuda')
bufs = dict(x1=x1, s1=s1, s2=s2, t=t, ta=ta, tb=tb, tc=tc)


def between(sync):
    if sync == 'event':
        e = torch.cuda.Event(); e.record(side); side.wait_event(e)
    elif sync == 'flush':
        with torch.cuda.stream(side):
            tiny.add_(1.0)


def run(src, busy, sync, chain, use_side=True):
    ss = side if use_side else main
    s = ss.cuda_stream
    if use_side:
        side.wait_stream(main)
    if chain == 'torchchain':
        with torch.cuda.stream(ss):
            torch.mul(src[:, 128:320], 2.0, out=ta); between(sync) if use_side else None
            torch.add(ta, 1.0, out=tb); between(sync) if use_side else None
            torch.mul(tb, tb, out=tc)
    else:
        lib.conv2d(L.view(src, 128, 384), sp['w1'], L.view(x1), B, 1, bias=sp['b1'], act=L.ACT_RELU, precision=P, stream=s)
        if use_side: between(sync)
        lib.rfft2(L.view(x1), L.view(s1), B, None, s)
        if use_side: between(sync)
        if chain == 'full':
            lib.conv2d(L.view(s1), fuw, L.view(s2), B, 1, bias=fub, act=L.ACT_RELU, precision=P, stream=s)
            if use_side: between(sync)
            lib.irfft2(L.view(s2), L.view(x1), L.view(t), B, None, s)
        else:
            lib.irfft2(L.view(s1), L.view(x1), L.view(t), B, None, s)
    if busy == 'conv':
        lib.conv2d(L.view(src), pk['w_lout'], L.view(dst, 0, 128), B, 3, 1, 1, L.PAD_REFLECT, False, pk['b_l'], L.ACT_RELU, None,
                   precision=P, stream=main.cuda_stream)
    elif busy == 'torch':
        torch.mul(big, 1.0001, out=big2)
    if use_side:
        main.wait_stream(side)


for variant in sys.argv[2:] or ['conv:none:full']:
    busy, sync, chain = variant.split(':')
    refs = []
    for src in srcs:
        for b_ in bufs.values(): b_.fill_(SENT)
        run(src, 'none', 'none', chain, use_side=False)
        torch.cuda.synchronize()
        refs.append({k: v.clone() for k, v in bufs.items()})
    bad, first = 0, None
    for it in range(N):
        k = it % 3
        for b_ in bufs.values(): b_.fill_(SENT)
        run(srcs[k], busy, sync, chain)
        torch.cuda.synchronize()
        wrong = [n for n in bufs if not torch.equal(bufs[n], refs[k][n])]
        if wrong:
            bad += 1
            if first is None:
                n0 = wrong[0]
                d = bufs[n0] != refs[k][n0]
                got = bufs[n0][d]
                prev = refs[(k + 2) % 3][n0][d]
                first = f'it {it}: {wrong}; {n0}: {int(d.sum())} elements, {int((got == SENT).sum())} sentinel, {int((got == prev).sum())} == previous iteration'
    print(f'== {variant}: {bad} mismatching iterations of {N}' + (f' [{first}]' if first else ''), flush=True)
