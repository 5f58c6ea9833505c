#!/usr/bin/env python3
"""CPU experiment for the next round: every stride-1 3x3 conv of the big-lama generator as Winograd F(2x2, 3x3) in the ORACLE (test
infrastructure; nothing here touches the HIP path) -- in fp32, and with the 3-term fp16 split applied to the TRANSFORMED operands (what an
MFMA implementation would multiply) -- against the direct fp32 oracle.   python tools/winograd_accuracy.py [res=256]"""
import sys, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import lama_oracle as O
torch.set_num_threads(8)
cfg = O.BIG_LAMA
sd = O.make_synthetic_state_dict(cfg, seed=0, calib_hw=64)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]])
Bt = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1.]])
At = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1.]])
def wino(x, w, split=False):
    # x already padded: [B,C,H+2,W+2], H,W even; w [Co,C,3,3]; fp32 Winograd F(2x2,3x3)
    Bn, C, Hp, Wp = x.shape
    H, W = Hp - 2, Wp - 2
    U = torch.einsum('ij,ocjk,lk->ocil', G, w, G)                     # [Co,C,4,4]
    t = x.unfold(2, 4, 2).unfold(3, 4, 2)                              # [B,C,H/2,W/2,4,4]
    V = torch.einsum('ij,bchwjk,lk->bchwil', Bt, t, Bt)
    if split:   # 3-term fp16 split of both operands (what the MFMA path would multiply), fp32 accumulate
        Uh = U.half().float(); Ul = (U - Uh).half().float()
        Vh = V.half().float(); Vl = (V - Vh).half().float()
        M = torch.einsum('ocil,bchwil->bohwil', Uh, Vh) + torch.einsum('ocil,bchwil->bohwil', Uh, Vl) + torch.einsum('ocil,bchwil->bohwil', Ul, Vh)
    else:
        M = torch.einsum('ocil,bchwil->bohwil', U, V)
    Y = torch.einsum('ij,bohwjk,lk->bohwil', At, M, At)                # [B,Co,H/2,W/2,2,2]
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(Bn, w.shape[0], H, W)
class WF(O._HalfOperands):
    split = False
    @staticmethod
    def conv2d(x, w, b=None, **kw):
        if w.shape[2] == 3 and kw.get('stride', 1) in (1, (1, 1)) and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0 and kw.get('padding', 0) in (0, (0, 0)) and kw.get('dilation', 1) in (1, (1, 1)) and kw.get('groups', 1) == 1:
            y = wino(x, w, WF.split)
            return y if b is None else y + b.view(1, -1, 1, 1)
        return torch.nn.functional.conv2d(x, w, b, **kw)
    @staticmethod
    def conv_transpose2d(x, w, b=None, **kw): return torch.nn.functional.conv_transpose2d(x, w, b, **kw)
hw, b = (int(sys.argv[1]) if len(sys.argv) > 1 else 256), 1
batch = O.make_synthetic_batch(b, hw, hw, seed=12)
x = torch.cat([batch['image'] * (1 - batch['mask']), batch['mask']], 1)
with torch.no_grad():
    ref = O.generator_forward(x, sd, cfg)
    ref64 = None
    for split in (False, True):
        WF.split = split
        keep = O.F; O.F = WF()
        try: y = O.generator_forward(x, sd, cfg)
        finally: O.F = keep
        print(hw, 'winograd F(2x2,3x3)', 'fp16x3 split' if split else 'fp32', 'max', float((y - ref).abs().max()), 'mean', float((y - ref).abs().mean()), flush=True)
