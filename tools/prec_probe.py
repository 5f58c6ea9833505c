#!/usr/bin/env python3
"""Error of one bottleneck conv (512->128, 3x3, K = 4608) against an fp64 reference: exact-fp32 MFMA kernel, bf16x3 kernel,
and the ideal 3-term split evaluated in fp64 (what the scheme would give with exact accumulation)."""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lama_amd import _lib as L  # noqa: E402

lib = L.get_lib()
g = torch.Generator().manual_seed(0)
B, cin, cout, H, W = 1, 512, 128, 64, 64
x = torch.relu(torch.randn(B, cin, H, W, generator=g)) * 3
w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
xp = F.pad(x.double(), (1, 1, 1, 1), mode='reflect')
ref = F.conv2d(xp, w.double())


def split(t, dt):
    h = t.to(dt).float()
    l = (t - h).to(dt).float()
    return h.double(), l.double()


for name, dt in (('bf16', torch.bfloat16), ('fp16', torch.float16)):
    xh, xl = split(x, dt)
    wh, wl = split(w, dt)
    pad = lambda t: F.pad(t, (1, 1, 1, 1), mode='reflect')
    y3 = F.conv2d(pad(xh), wh) + F.conv2d(pad(xl), wh) + F.conv2d(pad(xh), wl)
    print(f'ideal {name}x3 (fp64 accumulate): max {float((y3 - ref).abs().max()):.3e} mean {float((y3 - ref).abs().mean()):.3e}')
print(f'fp32 torch-CPU conv            : max {float((F.conv2d(xp.float(), w).double() - ref).abs().max()):.3e}')
st = torch.cuda.current_stream().cuda_stream
xd = x.cuda()
for name, prec in (('f32 kernel', L.PREC_F32), ('bf16x3 kernel', L.PREC_BF16X3), ('f16x3 kernel', L.PREC_F16X3)):
    wp = lib.pack_conv_weight(w.cuda(), None, precision=prec)
    y = torch.empty(B, cout, H, W, device='cuda')
    lib.conv2d(L.view(xd), wp, L.view(y), B, 3, 1, 1, L.PAD_REFLECT, False, None, L.ACT_NONE, precision=prec, stream=st)
    torch.cuda.synchronize()
    d = (y.cpu().double() - ref).abs()
    print(f'{name:30s} : max {float(d.max()):.3e} mean {float(d.mean()):.3e}   (|ref| mean {float(ref.abs().mean()):.2f})')
