#!/usr/bin/env python3
"""Per-workgroup timeline of the weights-in-registers kernels (LAMA_CW_TRACE).  usage: wr_trace.py [convA|fuconv|conv1]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
name = sys.argv[1] if len(sys.argv) > 1 else 'fuconv'
buf = torch.zeros(4096 * 32, dtype=torch.int64, device='cuda')
os.environ['LAMA_CW_TRACE'] = hex(buf.data_ptr())
import _toollib  # noqa: E402,F401  (LAMA_TOOL_LIB=<path>: another build of the library)
from lama_amd import _lib as L  # noqa: E402

lib = L.get_lib()
prec = L.PREC_F16X3
st = torch.cuda.current_stream().cuda_stream
B = 8
g = torch.Generator().manual_seed(0)
cin, cout, k, H, W = {'convA': (512, 128, 3, 64, 64), 'convB': (128, 384, 3, 64, 64), 'conv1': (384, 192, 1, 64, 64)}.get(name, (384, 384, 1, 64, 33))
x2c = 192 if name == 'convB' else 0      # the global-branch launch: 3x3 over x_l + fused 1x1 over t
x = torch.randn(B, cin, H, W, generator=g).cuda()
wt = torch.randn(cout, cin, k, k, generator=g).cuda()
wp = lib.pack_conv_weight(wt, None, stride=1, transposed=False, precision=prec)
y = torch.empty(B, cout, H, W, device='cuda')
bias = torch.randn(cout, generator=g).cuda()
x2 = torch.randn(B, x2c, H, W, generator=g).cuda() if x2c else None
w2p = lib.pack_conv_weight(torch.randn(cout, x2c, 1, 1, generator=g).cuda(), None, precision=prec) if x2c else None
resid = torch.randn(B, cout, H, W, generator=g).cuda() if x2c else None
for _ in range(5):
    lib.conv2d(L.view(x), wp, L.view(y), B, k, 1, k // 2, L.PAD_REFLECT, False, bias, L.ACT_RELU, None if resid is None else L.view(resid), None if x2 is None else L.view(x2), w2p, precision=prec, stream=st)
torch.cuda.synchronize()
buf.zero_()
torch.cuda.synchronize()
lib.conv2d(L.view(x), wp, L.view(y), B, k, 1, k // 2, L.PAD_REFLECT, False, bias, L.ACT_RELU, None if resid is None else L.view(resid), None if x2 is None else L.view(x2), w2p, precision=prec, stream=st)
torch.cuda.synchronize()
t = buf.view(-1, 32).cpu()
t = t[t[:, 0] > 0]
t0 = int(t[:, 0].min())
if name == 'convB':   # slots: 0 start, 1 first patch staged, 2..5 chunks of the 3x3 segment, 8 first patch of the 1x1 segment, 9..11 its chunks, 30 end
    rel = (t.double() - t0) / 100.0
    cols = [0, 1, 2, 3, 4, 5, 8, 9, 10, 11, 30]
    names = ['prologue', '3x3 chunk 0', 'chunk 1', 'chunk 2', 'chunk 3', '1x1 first patch', '1x1 chunk 0', 'chunk 1', 'chunk 2', 'epilogue']
    print(f'{name}: {t.shape[0]} workgroups (12 waves: 384 rows x 128 pixels); medians in us')
    for i, nm in enumerate(names):
        d = rel[:, cols[i + 1]] - rel[:, cols[i]]
        print(f'  {nm:18s} {d.median():6.2f}  (p90 {d.kthvalue(int(0.9 * d.numel())).values:6.2f})')
    print(f'  workgroup total    {(rel[:, 30] - rel[:, 0]).median():6.2f}; last end {rel[:, 30].max():.2f}')
    sys.exit(0)
nch = int(((t[:, 2:30] > 0).sum(1)).max())
print(f'{name}: {t.shape[0]} workgroups, {nch} chunks; times in us relative to the first workgroup start (100 MHz ticks)')
rel = (t.double() - t0) / 100.0
start, pro, end = rel[:, 0], rel[:, 1] - rel[:, 0], rel[:, 30] - rel[:, 0]
last = rel[:, 1 + nch]
print(f'start   : min {start.min():.2f} median {start.median():.2f} max {start.max():.2f}')
print(f'prologue: median {pro.median():.2f} max {pro.max():.2f}')
ch = (rel[:, 2:2 + nch] - rel[:, 1:1 + nch])
print('chunks  : median per chunk ' + ' '.join(f'{v:.2f}' for v in ch.median(0).values.tolist()))
print(f'epilogue: median {(rel[:, 30] - last).median():.2f} max {(rel[:, 30] - last).max():.2f}')
print(f'workgroup total: median {end.median():.2f} max {end.max():.2f}; last end {rel[:, 30].max():.2f}')
if name == 'convA' and int(t[:, 18].max()) > 0:   # shader-clock stamps of the k-steps of chunk 8 (wave 0 of every workgroup)
    ks = (t[:, 19:29] - t[:, 18:28]).double()
    print('chunk 8, k-steps 0..8 + barrier + stamp store (shader clocks, median over workgroups): ' + ' '.join(f'{v:.0f}' for v in ks.median(0).values.tolist()))
    tot = (t[:, 27] - t[:, 18]).double().median()
    print(f'chunk 8 k-steps total {tot:.0f} shader clocks; same chunk on the 100 MHz clock {ch[:, 8].median():.2f} us')
late = t[start > start.median() + 1.0].shape[0]
print(f'workgroups starting > 1 us after the median start (second round): {late}')
