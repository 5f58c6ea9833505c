#!/usr/bin/env python3
"""CPU experiment for the next round: what the generator's output costs when the two CROSS terms of the 3-term split run on the
fp8 matrix path.  ORACLE only (test infrastructure; nothing here touches the HIP path).

    w x  ~=  wh xh                (fp16 x fp16, fp32 accumulate: v_mfma_f32_32x32x16_f16, 1 unit of matrix-pipe time)
           + q8(wh) q8(xl)        (fp8 x fp8, fp32 accumulate: v_mfma_scale_f32_32x32x64_f8f6f4, 1/2 unit)
           + q8(wl) q8(xh)        (the same)                         -> 2 instead of 3 units per product

wh = half(w), wl = half(w - wh) (same for x); q8 = rounding to e4m3 (or e5m2) after a power-of-two scale -- per tensor (pessimistic) or per block of 32
along K (what the instruction's e8m0 block scales offer).  The cross terms are 2^-11 of the product, so ~4 significant bits each keep ~15 bits overall.
Every conv / transposed conv / spectral 1x1 of oracle.generator_forward is computed this way; result against the plain fp32 oracle.
    python tools/fp8_cross_terms_accuracy.py [res=256] [batch=1]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import lama_oracle as O
torch.set_num_threads(8)
Fn = torch.nn.functional


def q8(t, fmt, block_dim=None):
    """round to fp8 after a power-of-two scale that brings max |t| (per tensor, or per block of 32 along block_dim) to the top of the format's range"""
    top = 448.0 if fmt == torch.float8_e4m3fn else 57344.0
    if block_dim is None:
        m = t.abs().max().clamp_min(1e-30)
        s = torch.exp2(torch.floor(torch.log2(top / m)))
        return (t * s).to(fmt).float() / s
    # blocks of 32 along block_dim (K = input channels): the layout of the MX scale operands
    td = t.movedim(block_dim, -1)
    shp = td.shape
    K = shp[-1]
    pad = (-K) % 32
    if pad:
        td = Fn.pad(td, (0, pad))
    tb = td.reshape(*shp[:-1], -1, 32)
    m = tb.abs().amax(-1, keepdim=True).clamp_min(1e-30)
    s = torch.exp2(torch.floor(torch.log2(top / m)))
    out = ((tb * s).to(fmt).float() / s).reshape(*shp[:-1], -1)[..., :K]
    return out.movedim(-1, block_dim)


class Cross8:
    mode = 'f16x3'       # 'f16x3' (the shipped arithmetic), 'e4m3', 'e5m2', 'e4m3_block', 'hi_only'

    def __getattr__(self, name):
        return getattr(Fn, name)

    def _terms(self, x, w, op, kw, wk_dim):
        xh = x.half().float(); xl = (x - xh).half().float()
        wh = w.half().float(); wl = (w - wh).half().float()
        y = op(xh, wh, None, **kw)
        m = Cross8.mode
        if m == 'hi_only':
            return y
        if m == 'f16x3':
            return y + op(xl, wh, None, **kw) + op(xh, wl, None, **kw)
        fmt = torch.float8_e5m2 if m == 'e5m2' else torch.float8_e4m3fn
        blk = m.endswith('_block')
        qx = lambda t: q8(t, fmt, 1 if blk else None)          # activations: blocks along the channel axis
        qw = lambda t: q8(t, fmt, wk_dim if blk else None)
        return y + op(qx(xl), qw(wh), None, **kw) + op(qx(xh), qw(wl), None, **kw)

    def conv2d(self, x, w, b=None, **kw):
        y = self._terms(x, w, Fn.conv2d, kw, 1)
        return y if b is None else y + b.view(1, -1, 1, 1)

    def conv_transpose2d(self, x, w, b=None, **kw):
        y = self._terms(x, w, Fn.conv_transpose2d, kw, 0)
        return y if b is None else y + b.view(1, -1, 1, 1)


def main():
    res = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    bn = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    cfg = O.BIG_LAMA
    sd = O.make_synthetic_state_dict(cfg, seed=0, calib_hw=64)
    batch = O.make_synthetic_batch(bn, res, res, seed=12)
    x = torch.cat([batch['image'] * (1 - batch['mask']), batch['mask']], 1)
    with torch.no_grad():
        ref = O.generator_forward(x, sd, cfg)
        for mode in ('f16x3', 'e4m3_block', 'e4m3', 'e5m2', 'hi_only'):
            Cross8.mode = mode
            keep = O.F
            O.F = Cross8()
            try:
                y = O.generator_forward(x, sd, cfg)
            finally:
                O.F = keep
            d = (y - ref).abs()
            print(f'{bn} x {res}^2  cross terms {mode:11s}: max-abs {float(d.max()):.2e}  mean-abs {float(d.mean()):.2e}', flush=True)


if __name__ == '__main__':
    main()
