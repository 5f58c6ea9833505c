#!/usr/bin/env python3
"""Per-workgroup timeline of the weights-stationary pointwise GEMM (LAMA_GW_TRACE).  usage: gw_trace.py [conv1|fuconv] [nrot]
nrot > 1 rotates that many input / output buffer pairs so the operands do not sit in the 256 MiB Infinity Cache."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
name = sys.argv[1] if len(sys.argv) > 1 else 'fuconv'
nrot = int(sys.argv[2]) if len(sys.argv) > 2 else 6
buf = torch.zeros(512 * 16, dtype=torch.int64, device='cuda')
os.environ['LAMA_GW_TRACE'] = hex(buf.data_ptr())
from lama_amd import _lib as L  # noqa: E402

lib = L.get_lib()
prec = L.PREC_F16X3
st = torch.cuda.current_stream().cuda_stream
B = 8
g = torch.Generator().manual_seed(0)
cin, cout, H, W = (384, 192, 64, 64) if name == 'conv1' else (384, 384, 64, 33)
xs = [torch.randn(B, cin, H, W, generator=g).cuda() for _ in range(nrot)]
ys = [torch.empty(B, cout, H, W, device='cuda') for _ in range(nrot)]
wt = torch.randn(cout, cin, 1, 1, generator=g).cuda()
wp = lib.pack_conv_weight(wt, None, stride=1, transposed=False, precision=prec)
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
print(f'start: median {rel[:, 0].median():.2f} max {rel[:, 0].max():.2f};  last end {rel[:, 15].max():.2f}')
print(f'prologue (weights + first chunks): median {(rel[:, 1] - rel[:, 0]).median():.2f}')
print(f'tile 0 K loop: median {(rel[:, 14] - rel[:, 1]).median():.2f};  tile 0 epilogue: median {(rel[:, 2] - rel[:, 14]).median():.2f}')
nt = int((t[:, 2:8] > 0).sum(1).max())
prev = rel[:, 1]
for i in range(nt):
    cur = rel[:, 2 + i]
    ok = t[:, 2 + i] > 0
    print(f'  tile {i}: {int(ok.sum())} workgroups, median {(cur - prev)[ok].median():.2f} p90 {(cur - prev)[ok].quantile(0.9):.2f}')
    prev = cur
tot = rel[:, 15] - rel[:, 0]
print(f'workgroup total: median {tot.median():.2f} max {tot.max():.2f}')
ok = t[:, 13] > 0
seg = [('K loop (thread 0)', 2, 13), ('exchange writes', 13, 8), ('own sum + bias/resid requests', 8, 9), ('barrier 1', 9, 10), ('exchange reads', 10, 11), ('barrier 2', 11, 12), ('stores issued', 12, 3)]
print('tile 1, thread 0 (us, medians): ' + ' | '.join(f'{n} {(rel[:, b] - rel[:, a])[ok].median():.2f}' for n, a, b in seg))
