#!/usr/bin/env python3
"""VERDICT r1 "done" criterion for the co-residency hazard: N overlapped runs of one bottleneck FFC layer (spectral branch on a side
stream next to the local 3x3 conv) must be bit-identical to the serial run.  Mismatches are counted on the device (no host sync
per iteration).    python tools/overlap_stress.py [N=100000] [H=64]"""
import sys
import time
import torch
sys.path.insert(0, '.')
import torch.nn as nn
from lama_amd import ffc as F

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
H = W = int(sys.argv[2]) if len(sys.argv) > 2 else 64
B = 8
torch.manual_seed(0)
lay = F.FFC_BN_ACT(512, 512, kernel_size=3, ratio_gin=0.75, ratio_gout=0.75, padding=1, norm_layer=nn.BatchNorm2d,
                   activation_layer=nn.ReLU, enable_lfu=False).cuda()
lay.train(False)
for m in lay.modules():
    if isinstance(m, nn.BatchNorm2d):
        m.running_var.data.uniform_(0.5, 1.5); m.running_mean.data.normal_(0, 0.1)
srcs = [torch.randn(B, 512, H, W, device='cuda') for _ in range(2)]
resid = torch.randn(B, 512, H, W, device='cuda')
dst = torch.zeros_like(resid)
scratch = lay.make_scratch(srcs[0].shape, 'cuda')
F._DEFAULT_EXEC.cooperative_serial = True      # the overlapped order launches the local conv with LAMA_CONV_COOPERATIVE: same kernel geometry in the serial reference
refs = []
for s in srcs:
    lay.run(s, dst, scratch, resid)
    torch.cuda.synchronize()
    refs.append((dst.clone(), scratch['t'].clone(), scratch['x1'].clone()))
side = torch.cuda.Stream()
bad = torch.zeros(1, dtype=torch.int64, device='cuda')
t0 = time.time()
for it in range(N):
    k = it & 1
    dst.fill_(3.0); scratch['t'].fill_(3.0); scratch['x1'].fill_(3.0)
    lay.run(srcs[k], dst, scratch, resid, side=side)
    bad += (~torch.eq(dst, refs[k][0]).all() | ~torch.eq(scratch['t'], refs[k][1]).all() | ~torch.eq(scratch['x1'], refs[k][2]).all()).long()
    if it % 20000 == 19999:
        print(f'  {it + 1} runs, {int(bad)} mismatching, {time.time() - t0:.0f} s', flush=True)
torch.cuda.synchronize()
print(f'== overlap stress H={H} B={B}: {int(bad)} mismatching layer runs of {N} ({time.time() - t0:.0f} s)', flush=True)
