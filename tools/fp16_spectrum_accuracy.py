#!/usr/bin/env python3
"""CPU experiment (ORACLE only, test infrastructure): what the generator's output costs when the two spectra of every FourierUnit are STORED
as single fp16 values (Next #2 i):

    s1 = fp16(rfft2(x))                              the forward FFT rounds its output once (26 -> 13 MB at 8 x 512^2)
    s2 = fp16(relu((wh + wl) s1 + shift))            spectral 1x1 with hi + lo fp16 weights (BN scale folded before the split):
                                                     TWO MFMA products per MAC, no operand split, fp32 accumulate
    y  = irfft2(float(s2))                           fp32 arithmetic inside both FFTs

Everything else of the generator runs in fp32 ('rest=f32': isolates the effect) or in the shipped 3-term fp16 split ('rest=f16x3': what the product
would then be end to end).  Variants: both spectra fp16, only s1, only s2, and 'bf16' for comparison.
    python tools/fp16_spectrum_accuracy.py [res=512] [batch=1]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import lama_oracle as O
torch.set_num_threads(int(os.environ.get('OMP_NUM_THREADS', '8')))
Fn = torch.nn.functional


class Split3:
    """the shipped arithmetic of every conv: 3-term fp16 split, fp32 accumulate"""
    def __getattr__(self, name):
        return getattr(Fn, name)

    @staticmethod
    def _t(x, w, op, kw):
        xh = x.half().float(); xl = (x - xh).half().float()
        wh = w.half().float(); wl = (w - wh).half().float()
        return op(xh, wh, None, **kw) + op(xl, wh, None, **kw) + op(xh, wl, None, **kw)

    def conv2d(self, x, w, b=None, **kw):
        y = self._t(x, w, Fn.conv2d, kw)
        return y if b is None else y + b.view(1, -1, 1, 1)

    def conv_transpose2d(self, x, w, b=None, **kw):
        y = self._t(x, w, Fn.conv_transpose2d, kw)
        return y if b is None else y + b.view(1, -1, 1, 1)


STATS = {'s1max': 0.0, 's2max': 0.0}


def make_fu(mode_in, mode_out):
    def rnd(t, mode):
        if mode.startswith('f16lo'):          # fp16 everywhere except the |k| < T corner of low frequencies, which stays fp32
            T = int(mode[5:])
            h, wf = t.shape[-2], t.shape[-1]
            ky = torch.arange(h); ky = torch.minimum(ky, h - ky)
            keep = (ky[:, None] < T) & (torch.arange(wf)[None, :] < T)
            return torch.where(keep, t, t.half().float())
        if mode == 'f16':
            return t.half().float()
        if mode == 'bf16':
            return t.bfloat16().float()
        return t

    def fourier_unit(x, sd, p, calib=None):
        b, c, h, w = x.shape
        ff = torch.fft.rfftn(x, dim=(-2, -1), norm='ortho')
        ff = torch.stack((ff.real, ff.imag), dim=-1).permute(0, 1, 4, 2, 3).contiguous().view(b, -1, h, w // 2 + 1)
        STATS['s1max'] = max(STATS['s1max'], float(ff.abs().max()))
        ff = rnd(ff, mode_in)
        scale = sd[p + '.bn.weight'] / torch.sqrt(sd[p + '.bn.running_var'] + O.BN_EPS)
        shift = sd[p + '.bn.bias'] - sd[p + '.bn.running_mean'] * scale
        wf = sd[p + '.conv_layer.weight'] * scale[:, None, None, None]
        if mode_in == 'f32' or mode_in.startswith('f16lo'):
            y = Fn.conv2d(ff, wf)
        else:
            wh = wf.half().float(); wl = (wf - wh).half().float()
            y = Fn.conv2d(ff, wh) + Fn.conv2d(ff, wl)                       # two products, operand already 16-bit
        y = torch.relu(y + shift.view(1, -1, 1, 1))
        STATS['s2max'] = max(STATS['s2max'], float(y.abs().max()))
        y = rnd(y, mode_out)
        y = y.view(b, -1, 2, h, w // 2 + 1).permute(0, 1, 3, 4, 2).contiguous()
        return torch.fft.irfftn(torch.complex(y[..., 0], y[..., 1]), s=(h, w), dim=(-2, -1), norm='ortho')
    return fourier_unit


def main():
    res = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    bn = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    cfg = O.BIG_LAMA
    sd = O.make_synthetic_state_dict(cfg, seed=0, calib_hw=64)
    batch = O.make_synthetic_batch(bn, res, res, seed=12)
    x = torch.cat([batch['image'] * (1 - batch['mask']), batch['mask']], 1)
    keep_fu, keep_F = O.fourier_unit, O.F
    with torch.no_grad():
        ref = O.generator_forward(x, sd, cfg)
        variants = [('f32', 'f32'), ('f16', 'f16'), ('f16', 'f32'), ('f32', 'f16'), ('bf16', 'bf16')]
        if os.environ.get('TWOPRODUCT'):     # round 5: only the question of VERDICT r4 Next #2 -- the FourierUnit GEMM on two products (= fp16-stored first spectrum)
            variants = [('f16', 'f32'), ('f32', 'f16')]
        if os.environ.get('LOWBINS'):
            variants = [(f'f16lo{T}', f'f16lo{T}') for T in (1, 2, 4, 8, 16)]
        for rest in (('f32',) if os.environ.get('LOWBINS') else (('f16x3',) if os.environ.get('TWOPRODUCT') else ('f32', 'f16x3'))):
            for mi, mo in variants:
                if rest == 'f32' and (mi, mo) == ('f32', 'f32'):
                    continue
                STATS.update(s1max=0.0, s2max=0.0)
                O.fourier_unit = make_fu(mi, mo)
                O.F = Split3() if rest == 'f16x3' else keep_F
                try:
                    y = O.generator_forward(x, sd, cfg)
                finally:
                    O.fourier_unit, O.F = keep_fu, keep_F
                d = (y - ref).abs()
                print(f'{bn} x {res}^2  rest={rest:5s} spectrum in={mi:7s} out={mo:7s}: max-abs {float(d.max()):.2e}  mean-abs {float(d.mean()):.2e}'
                      f'   (max |s1| {STATS["s1max"]:.1f}, max |s2| {STATS["s2max"]:.1f})', flush=True)


if __name__ == '__main__':
    main()
