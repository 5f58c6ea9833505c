#!/usr/bin/env python3
"""Third round (see race_probe5/6.py): does a side-stream consumer READ wrong data (a torch copy of its input taken on the side
stream right before it would then be wrong too) or COMPUTE wrong output from right input (copy right, output wrong)?
Chain on the side stream with a snapshot copy of every stage's input; local 3x3 conv on the main stream.
Environment (profiling build via LAMA_HIP_LIB=lama_amd/lib/liblama_hip_prof.so): LAMA_CONV_WR=0 -> LDS-staged conv kernel on the main
stream, LAMA_FFT_INPLACE=0 -> two-buffer FFT kernels, LAMA_GEMM_WS=0 -> per-tile pointwise GEMMs."""
import sys
import torch
sys.path.insert(0, '.')
import torch.nn as nn
from lama_amd import ffc as F, _lib as L

lib = L.get_lib()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
tag = sys.argv[2] if len(sys.argv) > 2 else 'default'
H = W = 64
B = 8
wf = W // 2 + 1
torch.manual_seed(0)
lay = F.FFC_BN_ACT(512, 512, kernel_size=3, ratio_gin=0.75, ratio_gout=0.75, padding=1, norm_layer=nn.BatchNorm2d,
                   activation_layer=nn.ReLU, enable_lfu=False).cuda()
lay.train(False)
pk = lay._pack(); st_ = lay.ffc.convg2g; sp = st_._packed; fuw, fub = st_.fu._pack()
P = lay.precision
SENT = 12345.0
srcs = [torch.randn(B, 512, H, W, device='cuda') for _ in range(3)]
dst = torch.empty(B, 512, H, W, device='cuda')
x1 = torch.empty(B, 192, H, W, device='cuda'); t = torch.empty_like(x1)
s1 = torch.empty(B, 384, H, wf, device='cuda'); s2 = torch.empty_like(s1)
c_x1 = torch.empty_like(x1); c_s1 = torch.empty_like(s1); c_s2 = torch.empty_like(s1)
main = torch.cuda.current_stream(); side = torch.cuda.Stream()
bufs = dict(x1=x1, c_x1=c_x1, s1=s1, c_s1=c_s1, s2=s2, c_s2=c_s2, t=t)


def run(src, use_side):
    ss = side if use_side else main
    s = ss.cuda_stream
    if use_side:
        side.wait_stream(main)
    lib.conv2d(L.view(src, 128, 384), sp['w1'], L.view(x1), B, 1, bias=sp['b1'], act=L.ACT_RELU, precision=P, stream=s)
    with torch.cuda.stream(ss): c_x1.copy_(x1)
    lib.rfft2(L.view(x1), L.view(s1), B, None, s)
    with torch.cuda.stream(ss): c_s1.copy_(s1)
    lib.conv2d(L.view(s1), fuw, L.view(s2), B, 1, bias=fub, act=L.ACT_RELU, precision=P, stream=s)
    with torch.cuda.stream(ss): c_s2.copy_(s2)
    lib.irfft2(L.view(s2), L.view(x1), L.view(t), B, None, s)
    lib.conv2d(L.view(src), pk['w_lout'], L.view(dst, 0, 128), B, 3, 1, 1, L.PAD_REFLECT, False, pk['b_l'], L.ACT_RELU, None,
               precision=P, stream=main.cuda_stream)
    if use_side:
        main.wait_stream(side)


refs = []
for src in srcs:
    for b_ in bufs.values(): b_.fill_(SENT)
    run(src, False)
    torch.cuda.synchronize()
    refs.append({k: v.clone() for k, v in bufs.items()})
stats = {}
bad = 0
for it in range(N):
    k = it % 3
    for b_ in bufs.values(): b_.fill_(SENT)
    run(srcs[k], True)
    torch.cuda.synchronize()
    wrong = tuple(n for n in bufs if not torch.equal(bufs[n], refs[k][n]))
    if wrong:
        bad += 1
        stats[wrong] = stats.get(wrong, 0) + 1
print(f'== {tag}: {bad} mismatching iterations of {N}; patterns (which buffers differ from the serial run): '
      + ', '.join(f'{"+".join(k)}: {v}' for k, v in sorted(stats.items(), key=lambda kv: -kv[1])[:8]), flush=True)
