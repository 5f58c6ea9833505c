#!/usr/bin/env python3
"""VERDICT r5 Next #5 -- what a Winograd F(2x2, 3x3) form of the GLOBAL-branch launch (convl2g 128 -> 384 + conv2(t) 192 -> 384 + BN + ReLU + residual,
ffc.py:188-196,223, today ONE launch with the next layer's conv1 in its epilogue) can cost, from the building blocks that exist, at BASELINE
configs[1] (8 x 64 x 64 planes), operands rotated out of the Infinity Cache:

  A  today's fused launch (direct 3x3 + 1x1 + next conv1 in the epilogue) and the unfused one + stand-alone conv1
  B  the xi-row Winograd GEMM of wino_dev.inc at 128 -> 384 (three 128-row groups, K = 128 = four chunks), GEMM launch only (LAMA_CONV_DEFER_OUT):
     the transform-domain matrix work + staging a Winograd global kernel has to do whatever its epilogue -- a LOWER bound of any design whose
     workgroup holds one 128-row group (16 accumulator fragments per wave is the register file)
  C  B + its output transform over 384 channels (the partial sums through HBM: the design the verdict already rules out)
  D  the 1x1 conv2(t) 192 -> 384 as a launch of its own, and conv1 384 -> 192 as a launch of its own (a 128-row-group workgroup cannot feed the
     fused conv1: it needs all 384 rows of a pixel)

prints medians of HIP-event pairs."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _toollib  # noqa: E402,F401
from lama_amd import _lib as L  # noqa: E402


def timeit(fn, iters=30, warm=4):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    lib = L.get_lib()
    prec = L.PREC_F16X3
    dev = 'cuda'
    st = torch.cuda.current_stream().cuda_stream
    B, H, W, nrot = 8, 64, 64, 6
    g = torch.Generator().manual_seed(0)
    rnd = lambda *s: torch.randn(*s, generator=g).to(dev)          # noqa: E731
    xl = [rnd(B, 128, H, W) for _ in range(nrot)]
    t = [rnd(B, 192, H, W) for _ in range(nrot)]
    yg = [torch.empty(B, 384, H, W, device=dev) for _ in range(nrot)]
    x1 = [torch.empty(B, 192, H, W, device=dev) for _ in range(nrot)]
    res = [rnd(B, 384, H, W) for _ in range(nrot)]
    w3 = rnd(384, 128, 3, 3) * 0.03
    w1 = rnd(384, 192, 1, 1) * 0.05
    wc1 = rnd(192, 384, 1, 1) * 0.05
    bias, bias1 = rnd(384), rnd(192)
    wp3 = lib.pack_conv_weight(w3, None, precision=prec)
    wp1 = lib.pack_conv_weight(w1, None, precision=prec)
    wpc1 = lib.pack_conv_weight(wc1, None, precision=prec)
    cnt = [0]

    def rot(f):
        def run():
            i = cnt[0] % nrot
            cnt[0] += 1
            f(i)
        return run

    out = {}
    out['A0 direct 3x3 + 1x1 (unfused global launch)'] = timeit(rot(lambda i: lib.conv2d(
        L.view(xl[i]), wp3, L.view(yg[i]), B, 3, 1, 1, L.PAD_REFLECT, False, bias, L.ACT_RELU, L.view(res[i]), L.view(t[i]), wp1, precision=prec, stream=st)))
    out['D1 conv1 384 -> 192 as its own launch'] = timeit(rot(lambda i: lib.conv2d(
        L.view(yg[i]), wpc1, L.view(x1[i]), B, 1, 1, 0, L.PAD_ZERO, False, bias1, L.ACT_RELU, None, precision=prec, stream=st)))
    out['D2 conv2(t) 192 -> 384 as its own launch'] = timeit(rot(lambda i: lib.conv2d(
        L.view(t[i]), wp1, L.view(yg[i]), B, 1, 1, 0, L.PAD_ZERO, False, None, L.ACT_NONE, None, precision=prec, stream=st)))
    wpw = lib.pack_winograd_weight(w3, None, prec)
    ws = torch.zeros(lib.winograd_workspace_bytes(B, 384, H, W) // 4, device=dev)
    out['B  Winograd xi-row GEMM 128 -> 384, GEMM launch only'] = timeit(rot(lambda i: lib.winograd_conv3x3(
        L.view(xl[i]), wpw, L.view(yg[i]), B, ws, bias, L.ACT_RELU, L.view(res[i]), precision=prec, stream=st, defer_out=True)))
    out['C  ... + its output transform over 384 channels'] = timeit(rot(lambda i: lib.winograd_conv3x3(
        L.view(xl[i]), wpw, L.view(yg[i]), B, ws, bias, L.ACT_RELU, L.view(res[i]), precision=prec, stream=st)))
    for k, v in out.items():
        print(f'{k:62s} {v:8.1f} us')
    a0, b, c, d1, d2 = (out[k] for k in sorted(out))
    print(f'today: fused global launch 108-111 us (bench kernels_us) = unfused {a0:.1f} + conv1 in its epilogue ~13-19')
    print(f'lower bound of a 128-row-group Winograd global kernel + what it cannot fuse: B + conv1 = {b + d1:.1f} us (before its inverse transform, the 1x1 k-steps '
          f'of conv2(t) -- {d2:.1f} us as a launch -- and its epilogue)')
    print(f'the partial-sum design: C + D2 + D1 = {c + d2 + d1:.1f} us')


if __name__ == '__main__':
    main()
