#!/usr/bin/env python3
"""Hunt for the two-stream overlap race: run ONE bottleneck FFC_BN_ACT layer many times with the side stream and compare
every output / scratch buffer with a serial reference run."""
import sys
import torch
sys.path.insert(0, '.')
import torch.nn as nn
from lama_amd import ffc as F

torch.manual_seed(0)
B, H, W = int(sys.argv[1]) if len(sys.argv) > 1 else 4, 32, 32
lay = F.FFC_BN_ACT(512, 512, kernel_size=3, ratio_gin=0.75, ratio_gout=0.75, padding=1, norm_layer=nn.BatchNorm2d,
                   activation_layer=nn.ReLU, enable_lfu=False).cuda()
lay.train(False)
for m in lay.modules():
    if isinstance(m, nn.BatchNorm2d):
        m.running_var.data.uniform_(0.5, 1.5); m.running_mean.data.normal_(0, 0.1)
src = torch.randn(B, 512, H, W, device='cuda')
resid = torch.randn(B, 512, H, W, device='cuda')
dst = torch.zeros_like(src)
scratch = lay.make_scratch(src.shape, 'cuda')
lay.run(src, dst, scratch, resid)
torch.cuda.synchronize()
ref = dict(dst=dst.clone(), t=scratch['t'].clone(), x1=scratch['x1'].clone(), ws=scratch['ws'].clone())
nspec = B * 384 * H * (W // 2 + 1)
nspec_pad = ((nspec * 4 + 255) // 256 * 256) // 4
side = torch.cuda.Stream()
bad = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 300):
    dst.zero_(); scratch['t'].zero_(); scratch['x1'].zero_(); scratch['ws'].zero_()
    lay.run(src, dst, scratch, resid, side=side)
    torch.cuda.synchronize()
    msg = []
    s1 = scratch['ws'][:nspec].view(B, 384, H, W // 2 + 1); s2 = scratch['ws'][nspec_pad:nspec_pad + nspec].view(B, 384, H, W // 2 + 1)
    r1 = ref['ws'][:nspec].view(B, 384, H, W // 2 + 1); r2 = ref['ws'][nspec_pad:nspec_pad + nspec].view(B, 384, H, W // 2 + 1)
    for k, v in (('dst_l', dst[:, :128]), ('dst_g', dst[:, 128:]), ('t', scratch['t']), ('x1', scratch['x1']), ('s1', s1), ('s2', s2)):
        r = ref['dst'][:, :128] if k == 'dst_l' else ref['dst'][:, 128:] if k == 'dst_g' else r1 if k == 's1' else r2 if k == 's2' else ref[k]
        if not torch.equal(v, r):
            d = (v - r).abs()
            idx = (d > 0).nonzero()
            msg.append(f'{k}: max {float(d.max()):.3e} n={idx.shape[0]} first={idx[0].tolist()} last={idx[-1].tolist()}')
    if msg:
        bad += 1
        print(it, ' | '.join(msg), flush=True)
print('iterations with mismatch:', bad)
