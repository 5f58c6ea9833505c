#!/usr/bin/env python3
"""Root-causing the two-stream co-residency hazard: run the spectral branch of ONE bottleneck FFC layer on a side
stream next to the local 3x3 conv on the main stream, many times, and for every mismatching iteration report WHICH buffer went
wrong first, WHERE (image, channel / plane, rows), and WHAT the wrong values are (zero = read before written / lost store, equal
to the value of another iteration's input = stale cache line, or garbage).

    python tools/race_probe5.py <iterations> <H> [variant ...]
variants: full (default) | nolocal (no concurrent conv) | serial (everything on one stream) | fftonly (rfft2 + irfft2 on the side,
conv on main) | gemmonly (spectral GEMM on the side, conv on main) | conv1only
Inputs are re-randomised every iteration (so a stale line shows as the PREVIOUS iteration's value), buffers are poisoned with
a sentinel instead of zero.
"""
import sys
import torch
sys.path.insert(0, '.')
import torch.nn as nn
from lama_amd import ffc as F, _lib as L

lib = L.get_lib()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
H = W = int(sys.argv[2]) if len(sys.argv) > 2 else 64
variants = sys.argv[3:] or ['full']
B = 8 if H >= 64 else 4
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
main = torch.cuda.current_stream(); side = torch.cuda.Stream()
bufs = dict(x1=x1, s1=s1, s2=s2, t=t, dst=dst)


def run(src, variant, use_side):
    ss = side if use_side else main
    s = ss.cuda_stream
    if use_side:
        side.wait_stream(main)
    if variant in ('full', 'nolocal', 'serial', 'conv1only'):
        lib.conv2d(L.view(src, 128, 384), sp['w1'], L.view(x1), B, 1, bias=sp['b1'], act=L.ACT_RELU, precision=P, stream=s)
    if variant in ('full', 'nolocal', 'serial', 'fftonly'):
        lib.rfft2(L.view(x1), L.view(s1), B, None, s)
    if variant in ('full', 'nolocal', 'serial', 'gemmonly'):
        lib.conv2d(L.view(s1), fuw, L.view(s2), B, 1, bias=fub, act=L.ACT_RELU, precision=P, stream=s)
    if variant in ('full', 'nolocal', 'serial', 'fftonly'):
        lib.irfft2(L.view(s2), L.view(x1), L.view(t), B, None, s)
    if variant != 'nolocal':
        lib.conv2d(L.view(src), pk['w_lout'], L.view(dst, 0, 128), B, 3, 1, 1, L.PAD_REFLECT, False, pk['b_l'], L.ACT_RELU, None,
                   precision=P, stream=main.cuda_stream)
    if use_side:
        main.wait_stream(side)


def describe(name, got, ref, prev_ref):
    d = got != ref
    idx = d.nonzero()
    n = idx.shape[0]
    b0, c0 = int(idx[0][0]), int(idx[0][1])
    planes = torch.unique(idx[:, 0] * got.shape[1] + idx[:, 1])
    wrong = got[d]
    kinds = []
    if bool((wrong == SENT).all()): kinds.append('ALL SENTINEL (never written)')
    elif bool((wrong == SENT).any()): kinds.append(f'{int((wrong == SENT).sum())} sentinel')
    if prev_ref is not None and bool((wrong == prev_ref[d]).all()): kinds.append('== PREVIOUS iteration value (stale)')
    elif prev_ref is not None and bool((wrong == prev_ref[d]).any()): kinds.append(f'{int((wrong == prev_ref[d]).sum())} == previous iteration')
    if not bool(torch.isfinite(wrong).all()): kinds.append('non-finite')
    rows = torch.unique(idx[:, 2]).tolist()
    return (f'{name}: {n} elements in {planes.numel()} plane(s) [first (b={b0}, c={c0})], rows {rows[:6]}{"..." if len(rows) > 6 else ""} '
            f'max|diff| {float((got - ref)[d].abs().max()):.3e} {kinds}')


for variant in variants:
    # references per input, serial order
    refs = []
    for src in srcs:
        for b_ in bufs.values(): b_.fill_(SENT)
        run(src, 'serial' if variant == 'serial' else variant, False)
        torch.cuda.synchronize()
        refs.append({k: v.clone() for k, v in bufs.items()})
    bad = 0
    for it in range(N):
        k = it % 3
        for b_ in bufs.values(): b_.fill_(SENT)
        if variant in ('gemmonly',):
            s1.copy_(refs[k]['s1'])
        if variant in ('fftonly',):
            x1.copy_(refs[k]['x1']); s2.copy_(refs[k]['s2'])
        run(srcs[k], variant, variant != 'serial')
        torch.cuda.synchronize()
        msgs = []
        for name in ('x1', 's1', 's2', 't', 'dst'):
            if not torch.equal(bufs[name], refs[k][name]):
                msgs.append(describe(name, bufs[name], refs[k][name], refs[(k + 2) % 3][name] if it else None))
        if msgs:
            bad += 1
            if bad <= 8:
                print(f'[{variant} H={H}] it {it}: ' + ' | '.join(msgs), flush=True)
    print(f'== {variant} H={H} B={B}: {bad} mismatching iterations of {N}', flush=True)
