#!/usr/bin/env python3
"""Per-workgroup timeline of the super-tile pointwise GEMM (gemm1x1_w4_kernel, LAMA_GW_TRACE; profiling build).
usage: g4_trace.py [conv1|fuconv] [nrot]   (fuconv only: the traced instantiation is K = 384 without residual)
stamps (100 MHz): 0 start, 1 weights + first ring requested, 2 + u = K loop + exchange writes of super-tile u done (u < 10), 14 last epilogue done, 15 end"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
name = sys.argv[1] if len(sys.argv) > 1 else 'fuconv'
nrot = int(sys.argv[2]) if len(sys.argv) > 2 else 6
buf = torch.zeros(512 * 16, dtype=torch.int64, device='cuda')
os.environ['LAMA_GW_TRACE'] = hex(buf.data_ptr())
os.environ.setdefault('LAMA_HIP_LIB', os.path.join(ROOT, 'lama_amd', 'lib', 'liblama_hip_prof.so'))
from lama_amd import _lib as L  # noqa: E402

lib = L.get_lib()
prec = L.PREC_F16X3
st = torch.cuda.current_stream().cuda_stream
B = 8
g = torch.Generator().manual_seed(0)
cin, cout, H, W = (384, 192, 64, 64) if name == 'conv1' else (384, 384, 64, 33)
xs = [torch.randn(B, cin, H, W, generator=g).cuda() for _ in range(nrot)]
ys = [torch.empty(B, cout, H, W, device='cuda') for _ in range(nrot)]
wp = lib.pack_conv_weight(torch.randn(cout, cin, 1, 1, generator=g).cuda(), None, stride=1, transposed=False, precision=prec)
bias = torch.randn(cout, generator=g).cuda()
run = lambda i: lib.conv2d(L.view(xs[i % nrot]), wp, L.view(ys[i % nrot]), B, 1, 1, 0, L.PAD_REFLECT, False, bias, L.ACT_RELU, None, None, None, precision=prec, stream=st)
for i in range(2 * nrot):
    run(i)
torch.cuda.synchronize()
buf.zero_()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); run(0); b.record()
torch.cuda.synchronize()
t = buf.view(-1, 16).cpu()
t = t[t[:, 0] > 0]
t0 = int(t[:, 0].min())
rel = (t.double() - t0) / 100.0
print(f'{name} nrot={nrot}: {t.shape[0]} workgroups, event time {a.elapsed_time(b) * 1e3:.1f} us; us relative to the first start')
print(f'start: median {rel[:, 0].median():.2f} max {rel[:, 0].max():.2f};  end: median {rel[:, 15].median():.2f} last {rel[:, 15].max():.2f}')
print(f'prologue (first ring + weights requested): median {(rel[:, 1] - rel[:, 0]).median():.2f}')
prev = rel[:, 1]
for u in range(10):
    ok = t[:, 2 + u] > 0
    if not bool(ok.any()):
        break
    k = rel[:, 2 + u]
    print(f'  super-tile {u}: {int(ok.sum())} workgroups, K loop + exchange writes median {(k - prev)[ok].median():.2f} p90 {(k - prev)[ok].quantile(0.9):.2f} | done at median {k[ok].median():.2f} max {k[ok].max():.2f}')
    prev = k
print(f'last epilogue (bare): median {(rel[:, 14] - prev).median():.2f};  end: median {rel[:, 15].median():.2f} last {rel[:, 15].max():.2f}')
