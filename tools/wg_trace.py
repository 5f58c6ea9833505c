#!/usr/bin/env python3
"""Per-workgroup timeline of the Winograd local conv (wino_gemm_kernel, LAMA_WG_TRACE; profiling library).
usage: LAMA_TOOL_LIB=lama_amd/lib/liblama_hip_prof.so wg_trace.py [H=64] [nrot=4]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
H = int(sys.argv[1]) if len(sys.argv) > 1 else 64
nrot = int(sys.argv[2]) if len(sys.argv) > 2 else 4
buf = torch.zeros(2048 * 16, dtype=torch.int64, device='cuda')
os.environ['LAMA_WG_TRACE'] = hex(buf.data_ptr())
import _toollib  # noqa: E402,F401  (LAMA_TOOL_LIB=<path>: another build of the library)
from lama_amd import _lib as L  # noqa: E402

lib = L.get_lib()
prec = L.PREC_F16X3
st = torch.cuda.current_stream().cuda_stream
B = 8 if H == 64 else 2
g = torch.Generator().manual_seed(0)
xs = [torch.randn(B, 512, H, H, generator=g).cuda() for _ in range(nrot)]
ys = [torch.empty(B, 128, H, H, device='cuda') for _ in range(nrot)]
rs = [torch.randn(B, 128, H, H, generator=g).cuda() for _ in range(nrot)]
wp = lib.pack_winograd_weight(torch.randn(128, 512, 3, 3, generator=g).cuda() * 0.03, None, prec)
bias = torch.randn(128, generator=g).cuda()
ws = torch.empty(lib.winograd_workspace_bytes(B, 128, H, H), dtype=torch.uint8, device='cuda')
run = lambda i: lib.winograd_conv3x3(L.view(xs[i % nrot]), wp, L.view(ys[i % nrot]), B, ws, bias, L.ACT_RELU, L.view(rs[i % nrot]), precision=prec, stream=st)
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
print(f'winograd local conv {B} x 512 -> 128 at {H} x {H}, nrot={nrot}: {t.shape[0]} workgroups, event time (both launches) {a.elapsed_time(b) * 1e3:.1f} us')
print(f'start: median {rel[:, 0].median():.2f} max {rel[:, 0].max():.2f};  last end {rel[:, 3].max():.2f}')
print(f'prologue (first chunk transformed): median {(rel[:, 1] - rel[:, 0]).median():.2f} max {(rel[:, 1] - rel[:, 0]).max():.2f}')
print(f'K loop (16 chunks of 32 channels): median {(rel[:, 2] - rel[:, 1]).median():.2f} p90 {(rel[:, 2] - rel[:, 1]).quantile(0.9):.2f} max {(rel[:, 2] - rel[:, 1]).max():.2f}')
print(f'exchange + half inverse + stores: median {(rel[:, 3] - rel[:, 2]).median():.2f} max {(rel[:, 3] - rel[:, 2]).max():.2f}')
ck = t[:, 4:8].double()
print(f'chunk 8, wave 0, shader clocks: region 0 median {(ck[:, 1] - ck[:, 0]).median():.0f}, region 1 {(ck[:, 2] - ck[:, 1]).median():.0f}, barrier wait {(ck[:, 3] - ck[:, 2]).median():.0f}, chunk {(ck[:, 3] - ck[:, 0]).median():.0f} (48 MFMAs = 1536 clocks per region)')
