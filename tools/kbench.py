#!/usr/bin/env python3
"""Per-kernel timing at BASELINE config-2 shapes (B=8, 512x512 -> bottleneck 64x64).  Writes JSON to
gpurun_out/kbench.json.  Times are medians of HIP-event pairs on the launch stream."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _toollib  # noqa: E402,F401  (LAMA_TOOL_LIB=<path>: another build of the library)
from lama_amd import _lib as L  # noqa: E402


def timeit(fn, iters=20, warm=3):
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
    return ts[len(ts) // 2], ts[0]


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else 'f16x3'
    only_fu = len(sys.argv) > 2 and sys.argv[2] == 'fu'      # spectral branch only: the 1x1 GEMMs and the FFT kernels
    out_tag = sys.argv[3] if len(sys.argv) > 3 else tag
    # KBENCH_ROT=n: rotate n operand sets per case so that inputs / outputs do not sit in the 256 MiB Infinity Cache (pipeline conditions)
    nrot = int(os.environ.get('KBENCH_ROT', '1'))
    prec = L.PREC_NAMES[tag]
    lib = L.get_lib()
    dev = 'cuda'
    st = torch.cuda.current_stream().cuda_stream
    B, h, w = 8, 64, 64
    res = {}
    g = torch.Generator().manual_seed(0)

    def rnd(*s):
        return torch.randn(*s, generator=g).to(dev)

    def conv_case(name, cin, cout, k, H, W, stride=1, tr=False, x2c=0, flops=None, bytes_=None):
        if only_fu and k != 1:
            return
        x = rnd(B, cin, H, W)
        wt = rnd(cin, cout, k, k) if tr else rnd(cout, cin, k, k)
        stride = 2 if tr else stride
        wp = lib.pack_conv_weight(wt, None, stride=stride, transposed=tr, precision=prec)
        Ho, Wo = (2 * H, 2 * W) if tr else ((H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1)
        y = torch.empty(B, cout, Ho, Wo, device=dev)
        bias = rnd(cout)
        x2 = w2p = None
        if x2c:
            x2 = rnd(B, x2c, Ho, Wo); w2p = lib.pack_conv_weight(rnd(cout, x2c, 1, 1), None, precision=prec)
        xs = [x] + [torch.randn_like(x) for _ in range(nrot - 1)]
        ys = [y] + [torch.empty_like(y) for _ in range(nrot - 1)]
        cnt = [0]

        def fn():
            i = cnt[0] % nrot
            cnt[0] += 1
            lib.conv2d(L.view(xs[i]), wp, L.view(ys[i]), B, k, stride, 1 if tr else k // 2, L.PAD_ZERO if tr else L.PAD_REFLECT, tr, bias,
                       L.ACT_RELU, None, None if x2 is None else L.view(x2), w2p, precision=prec, stream=st)
        med, mn = timeit(fn)
        fl = 2.0 * B * Ho * Wo * cout * (cin * k * k / (4 if tr else 1) * (2.25 / 2.25 if not tr else 1) + x2c) if flops is None else flops
        if tr:
            fl = 2.0 * B * H * W * cout * cin * 9
        res[name] = dict(us=med, us_min=mn, tflops=fl / med / 1e6, gflop=fl / 1e9)
        print(name, res[name], flush=True)

    conv_case('conv3x3_512to128 (l2l+g2l)', 512, 128, 3, h, w)
    conv_case('conv3x3_128to384+1x1_192 (l2g+conv2)', 128, 384, 3, h, w, x2c=192)
    conv_case('conv1x1_384to192 (st.conv1)', 384, 192, 1, h, w)
    conv_case('conv1x1_384to384_spec (fu.conv)', 384, 384, 1, h, 33)
    conv_case('stem7x7_4to64', 4, 64, 7, 512, 512)
    conv_case('down3x3s2_64to128', 64, 128, 3, 512, 512, stride=2)
    conv_case('down3x3s2_128to256', 128, 256, 3, 256, 256, stride=2)
    conv_case('down3x3s2_256to512', 256, 512, 3, 128, 128, stride=2)
    conv_case('up_512to256', 512, 256, 3, 64, 64, tr=True)
    conv_case('up_256to128', 256, 128, 3, 128, 128, tr=True)
    conv_case('up_128to64', 128, 64, 3, 256, 256, tr=True)
    conv_case('head7x7_64to3', 64, 3, 7, 512, 512)

    # FFT kernels
    x1 = rnd(B, 192, h, w)
    spec = torch.empty(B, 384, h, w // 2 + 1, device=dev)
    y = torch.empty_like(x1)
    x1s = [x1] + [torch.randn_like(x1) for _ in range(nrot - 1)]
    specs = [spec] + [torch.randn_like(spec) for _ in range(nrot - 1)]
    ysr = [y] + [torch.empty_like(y) for _ in range(nrot - 1)]
    cnt = [0]

    def rot(f):
        def g():
            i = cnt[0] % nrot
            cnt[0] += 1
            f(i)
        return g
    med, mn = timeit(rot(lambda i: lib.rfft2(L.view(x1s[i]), L.view(specs[i]), B, None, st)))
    res['rfft2_8x192x64x64'] = dict(us=med, us_min=mn, gbps=(x1.numel() + spec.numel()) * 4 / med / 1e3)
    med, mn = timeit(rot(lambda i: lib.irfft2(L.view(specs[i]), L.view(x1s[i]), L.view(ysr[i]), B, None, st)))
    res['irfft2_add_8x192x64x64'] = dict(us=med, us_min=mn, gbps=(2 * x1.numel() + spec.numel()) * 4 / med / 1e3)
    wp = lib.pack_conv_weight(rnd(384, 384, 1, 1), None, precision=prec)
    bias = rnd(384)
    ws = torch.empty(lib.fourier_unit_workspace_bytes(B, 192, h, w) // 4 + 1, device=dev)
    med, mn = timeit(rot(lambda i: lib.fourier_unit(L.view(x1s[i]), wp, bias, L.view(ysr[i]), B, True, ws, precision=prec, stream=st)))
    alg = 2 * x1.numel() * 4 + 384 * 384 * 4 + 384 * 4
    res['fourier_unit_8x192x64x64'] = dict(us=med, us_min=mn, alg_bytes=alg, alg_gbps=alg / med / 1e3, frac_of_8TBs=alg / med / 1e3 / 8000)
    for k in ('rfft2_8x192x64x64', 'irfft2_add_8x192x64x64', 'fourier_unit_8x192x64x64'):
        print(k, res[k], flush=True)

    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    with open(os.path.join(ROOT, 'gpurun_out', f'kbench_{out_tag}.json'), 'w') as f:
        json.dump(res, f, indent=1)


if __name__ == '__main__':
    main()
