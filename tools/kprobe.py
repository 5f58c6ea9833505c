#!/usr/bin/env python3
"""Run selected bottleneck kernels a few times (for rocprofv3 --pmc passes).  usage: kprobe.py [bf16x3|f32] [names...]
names: convA convAc convB conv1 fuconv rfft irfft down1 up3 stem head"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _toollib  # noqa: E402,F401  (LAMA_TOOL_LIB=<path>: another build of the library)
from lama_amd import _lib as L  # noqa: E402


def main():
    prec = L.PREC_NAMES[sys.argv[1] if len(sys.argv) > 1 else 'f16x3']
    names = sys.argv[2:] or ['convA', 'convB', 'conv1', 'fuconv', 'rfft', 'irfft']
    iters = int(os.environ.get('KPROBE_ITERS', '10'))
    lib = L.get_lib()
    dev = 'cuda'
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(0)
    B, h, w = 8, 64, 64

    zero = os.environ.get('KPROBE_ZERO', '0') == '1'   # all-zero operands: how much of the time is power (operand toggling)?

    def rnd(*s):
        return torch.zeros(*s, device=dev) if zero else torch.randn(*s, generator=g).to(dev)

    def conv(cin, cout, k, H, W, stride=1, tr=False, x2c=0, fuse1=False, coop=False):
        x = rnd(B, cin, H, W)
        wt = rnd(cin, cout, k, k) if tr else rnd(cout, cin, k, k)
        s2 = 2 if tr else stride
        wp = lib.pack_conv_weight(wt, None, stride=s2, transposed=tr, precision=prec)
        Ho, Wo = (2 * H, 2 * W) if tr else ((H + 2 * (k // 2) - k) // s2 + 1, (W + 2 * (k // 2) - k) // s2 + 1)
        y = torch.empty(B, cout, Ho, Wo, device=dev)
        bias = rnd(cout)
        x2 = w2p = None
        if x2c:
            x2 = rnd(B, x2c, Ho, Wo)
            w2p = lib.pack_conv_weight(rnd(cout, x2c, 1, 1), None, precision=prec)
        f1 = None
        if fuse1:   # conv1 of the next layer in this launch's epilogue (lama_conv2d_args.fuse1_*)
            order = lib.fuse1_channel_order()
            w1f = lib.pack_conv_weight(rnd(192, cout, 1, 1)[:, order.to(dev)].contiguous(), None, precision=prec)
            x1buf = torch.empty(B, 192, Ho, Wo, device=dev)
            f1 = (w1f, rnd(192), L.view(x1buf), x1buf)          # the last entry keeps the buffer alive
        resid = rnd(B, cout, Ho, Wo) if fuse1 else None
        return lambda: lib.conv2d(L.view(x), wp, L.view(y), B, k, s2, 1 if tr else k // 2, L.PAD_ZERO if tr else L.PAD_REFLECT, tr, bias,
                                  L.ACT_RELU, None if resid is None else L.view(resid), None if x2 is None else L.view(x2), w2p, precision=prec,
                                  stream=st, fuse1=None if f1 is None else f1[:3], cooperative=coop)

    def wino(cin, cout, H, W):
        x = rnd(B, cin, H, W)
        wp = lib.pack_winograd_weight(rnd(cout, cin, 3, 3) * 0.03, None, prec)
        y = torch.empty(B, cout, H, W, device=dev)
        bias, resid = rnd(cout), rnd(B, cout, H, W)
        ws = torch.empty(lib.winograd_workspace_bytes(B, cout, H, W), dtype=torch.uint8, device=dev)
        return lambda: lib.winograd_conv3x3(L.view(x), wp, L.view(y), B, ws, bias, L.ACT_RELU, L.view(resid), precision=prec, stream=st)

    def fu_rot():
        # FourierUnit.forward (ffc.py:76-113) as lama_fourier_unit_fwd, KPROBE_ROT operand sets used round-robin so that a call does
        # not find its operands of the previous call in the 256 MiB Infinity Cache (default 6 sets x ~130 MB)
        nrot = int(os.environ.get('KPROBE_ROT', '6'))
        wp = lib.pack_conv_weight(rnd(384, 384, 1, 1) * 0.05, None, precision=prec)
        bias = rnd(384)
        sets = [(rnd(B, 192, h, w), torch.empty(B, 192, h, w, device=dev),
                 torch.empty(lib.fourier_unit_workspace_bytes(B, 192, h, w), dtype=torch.uint8, device=dev)) for _ in range(nrot)]
        k = [0]

        def run():
            x, yy, ws = sets[k[0] % nrot]
            k[0] += 1
            lib.fourier_unit(L.view(x), wp, bias, L.view(yy), B, True, ws, precision=prec, stream=st)
        return run

    x1 = rnd(B, 192, h, w)
    spec = torch.empty(B, 384, h, w // 2 + 1, device=dev)
    y = torch.empty_like(x1)
    table = {
        'convA': lambda: conv(512, 128, 3, h, w),
        'convAc': lambda: conv(512, 128, 3, h, w, coop=True),     # LAMA_CONV_COOPERATIVE: one 4-wave workgroup per CU (the geometry of the overlapped step)
        'convB': lambda: conv(128, 384, 3, h, w, x2c=192),
        'convBf': lambda: conv(128, 384, 3, h, w, x2c=192, fuse1=True),
        'convA128': lambda: conv(512, 128, 3, 128, 128),
        'convB128': lambda: conv(128, 384, 3, 128, 128, x2c=192),
        'conv1': lambda: conv(384, 192, 1, h, w),
        'fuconv': lambda: conv(384, 384, 1, h, 33),
        'down1': lambda: conv(64, 128, 3, 512, 512, stride=2),
        'down2': lambda: conv(128, 256, 3, 256, 256, stride=2),
        'down3': lambda: conv(256, 512, 3, 128, 128, stride=2),
        'up1': lambda: conv(512, 256, 3, 64, 64, tr=True),
        'up2': lambda: conv(256, 128, 3, 128, 128, tr=True),
        'up3': lambda: conv(128, 64, 3, 256, 256, tr=True),
        'stem': lambda: conv(4, 64, 7, 512, 512),
        'head': lambda: conv(64, 3, 7, 512, 512),
        'fu': lambda: fu_rot(),
        'wino': lambda: wino(512, 128, h, w),              # the local conv (convA) as Winograd F(2x2, 3x3): lama_winograd_conv3x3_fwd
        'wino128': lambda: wino(512, 128, 128, 128),
        'rfft': lambda: (lambda: lib.rfft2(L.view(x1), L.view(spec), B, None, st)),
        'irfft': lambda: (lambda: lib.irfft2(L.view(spec), L.view(x1), L.view(y), B, None, st)),
    }
    for n in names:
        fn = table[n]()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        print(n, round(a.elapsed_time(b) * 1e3 / iters, 2), 'us', flush=True)


if __name__ == '__main__':
    main()
