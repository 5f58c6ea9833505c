#!/usr/bin/env python3
"""Ad-hoc conv check on the GPU against torch-CPU: python tools/conv_check.py cin cout k stride pad H W [zero|reflect] ..."""
import sys
import torch
import torch.nn.functional as F
sys.path.insert(0, '.')
from lama_amd import _lib as L
lib = L.get_lib()
g = torch.Generator().manual_seed(0)
args = sys.argv[1:]
while args:
    cin, cout, k, stride, pad, H, W = map(int, args[:7]); mode = args[7]; args = args[8:]
    for prec in (L.PREC_F32, L.PREC_BF16X3, L.PREC_F16X3):
        x = torch.randn(1, cin, H, W, generator=g); w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
        xp = F.pad(x, (pad,) * 4) if mode == 'zero' else F.pad(x, (pad,) * 4, mode='reflect')
        ref = F.conv2d(xp, w, stride=stride)
        y = torch.zeros(ref.shape, device='cuda')
        xd = x.cuda()
        wp = lib.pack_conv_weight(w.cuda(), None, stride=stride, precision=prec)
        lib.conv2d(L.view(xd), wp, L.view(y), 1, k, stride, pad, L.PAD_ZERO if mode == 'zero' else L.PAD_REFLECT, False, None, L.ACT_NONE,
                   precision=prec, stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        d = (y.cpu() - ref).abs()
        bad = (d > 1e-3).nonzero()
        print((cin, cout, k, stride, pad, H, W, mode), 'prec', prec, 'err', float(d.max()), 'nbad', bad.shape[0],
              bad[0].tolist() if bad.shape[0] else '', bad[-1].tolist() if bad.shape[0] else '', flush=True)
