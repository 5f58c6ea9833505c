#!/usr/bin/env python3
"""Timings of the eight OUTER layers of the generator at BASELINE configs[1] shapes (B = 8, 512 x 512): stem, down1-3, up1-3, head -- each launched through
the C ABI on rotated operand sets (inputs / outputs do not sit in the Infinity Cache), medians of HIP-event pairs.  For same-box A/B runs of the
profiling library's switches:   LAMA_TOOL_LIB=lama_amd/lib/liblama_hip_prof.so LAMA_CT=3 python tools/outer_ab.py [names...]
Prints one line per layer: name, median us, min us, checksum of the output (so that two variants can be compared for equality)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _toollib  # noqa: E402,F401  (LAMA_TOOL_LIB=<path>: another build of the library)
from lama_amd import _lib as L  # noqa: E402

LAYERS = {           # name: (cin, cout, k, H, W, stride, transposed, pad)
    'stem': (4, 64, 7, 512, 512, 1, False, 3),
    'down1': (64, 128, 3, 512, 512, 2, False, 1),
    'down2': (128, 256, 3, 256, 256, 2, False, 1),
    'down3': (256, 512, 3, 128, 128, 2, False, 1),
    'up1': (512, 256, 3, 64, 64, 2, True, 1),
    'up2': (256, 128, 3, 128, 128, 2, True, 1),
    'up3': (128, 64, 3, 256, 256, 2, True, 1),
    'head': (64, 3, 7, 512, 512, 1, False, 3),
}


def main():
    names = sys.argv[1:] or list(LAYERS)
    prec = L.PREC_NAMES[os.environ.get('LAMA_PRECISION', 'f16x3')]
    lib = L.get_lib()
    dev, B, nrot = 'cuda', 8, int(os.environ.get('KBENCH_ROT', '3'))
    g = torch.Generator().manual_seed(0)
    tag = ' '.join(f'{k}={v}' for k, v in sorted(os.environ.items()) if k.startswith('LAMA_') and k != 'LAMA_TOOL_LIB')
    for name in names:
        cin, cout, k, H, W, stride, tr, pad = LAYERS[name]
        x = [torch.randn(B, cin, H, W, generator=g).to(dev) for _ in range(nrot)]
        wt = (torch.randn(cin, cout, k, k, generator=g) if tr else torch.randn(cout, cin, k, k, generator=g)).to(dev) * 0.05
        wp = lib.pack_conv_weight(wt, None, stride=stride, transposed=tr, precision=prec)
        Ho, Wo = (2 * H, 2 * W) if tr else ((H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1)
        y = [torch.empty(B, cout, Ho, Wo, device=dev) for _ in range(nrot)]
        bias = torch.randn(cout, generator=g).to(dev)
        act = L.ACT_SIGMOID if name == 'head' else L.ACT_RELU
        st = torch.cuda.current_stream().cuda_stream
        it = [0]

        def fn():
            i = it[0] % nrot
            it[0] += 1
            lib.conv2d(L.view(x[i]), wp, L.view(y[i]), B, k, stride, pad, L.PAD_ZERO if tr else L.PAD_REFLECT, tr, bias, act, None, precision=prec, stream=st)
        for _ in range(2 * nrot):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(15):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 1e3)
        ts.sort()
        it[0] = 0
        fn()
        torch.cuda.synchronize()
        cs = float(y[0].double().sum()), float(y[0].double().abs().sum())
        print(f'{name:6s} [{tag}] median {ts[len(ts) // 2]:8.1f} us  min {ts[0]:8.1f} us  checksum {cs[0]:.6e} {cs[1]:.6e}', flush=True)
        del x, y


if __name__ == '__main__':
    main()
