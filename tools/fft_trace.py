#!/usr/bin/env python3
"""Per-workgroup phase timeline of the 64x64 one-plane FFT kernels (LAMA_FFT_TRACE).  usage: fft_trace.py [rfft|irfft]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
name = sys.argv[1] if len(sys.argv) > 1 else 'rfft'
buf = torch.zeros(2048 * 16, dtype=torch.int64, device='cuda')
os.environ['LAMA_FFT_TRACE'] = hex(buf.data_ptr())
os.environ['LAMA_FFT_SEQ'] = '1'
from lama_amd import _lib as L  # noqa: E402

lib = L.get_lib()
st = torch.cuda.current_stream().cuda_stream
B, C, h, w = 8, 192, 64, 64
g = torch.Generator().manual_seed(0)
x = torch.randn(B, C, h, w, generator=g).cuda()
spec = torch.randn(B, 2 * C, h, w // 2 + 1, generator=g).cuda()
y = torch.empty_like(x)
fn = (lambda: lib.rfft2(L.view(x), L.view(spec), B, None, st)) if name == 'rfft' else (lambda: lib.irfft2(L.view(spec), L.view(x), L.view(y), B, None, st))
for _ in range(5):
    fn()
torch.cuda.synchronize()
buf.zero_()
torch.cuda.synchronize()
fn()
torch.cuda.synchronize()
t = buf.view(-1, 16).cpu()
t = t[t[:, 0] > 0]
t0 = int(t[:, 0].min())
rel = (t[:, :7].double() - t0) / 100.0
print(f'{name}: {t.shape[0]} workgroups; us relative to the first start (100 MHz ticks)')
print(f'start: min {rel[:, 0].min():.2f} median {rel[:, 0].median():.2f} max {rel[:, 0].max():.2f}; last end {rel[:, 6].max():.2f}')
names = ['twiddles+load', 'fft A', 'untangle', 'fft B', 'dc/pack', 'store'] if name == 'rfft' else ['twiddles+load', 'pack col0', 'col fft', 'row pairs', 'row fft', 'store']
d = rel[:, 1:7] - rel[:, 0:6]
for i, n in enumerate(names):
    print(f'  {n:14s} median {d[:, i].median():.2f}  p90 {d[:, i].quantile(0.9):.2f}')
tot = rel[:, 6] - rel[:, 0]
print(f'workgroup total: median {tot.median():.2f} p90 {tot.quantile(0.9):.2f} max {tot.max():.2f}')
first = rel[:, 0] < 1.0
print(f'first-round workgroups (start < 1 us): {int(first.sum())}; their median total {tot[first].median():.2f}')
