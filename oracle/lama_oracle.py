"""CPU oracle for the LaMa FFC hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this file.  ``lama_amd/`` never imports it; the product path has no CPU fallback.

What this is: a functional (state-dict driven) restatement, on torch-CPU fp32 ops, of the arithmetic
the reference executes for ``bin/predict.py`` -> ``FFCResNetGenerator`` (advimman/lama).  Every
function cites the reference ``file:line`` it follows (paths relative to the reference root).
The arithmetic primitives live in third-party PyTorch (``torch.fft.rfftn/irfftn``, ``conv2d``,
``batch_norm``, ``conv_transpose2d``; pinned by the reference at torch==1.8.x, this image ships
2.10.0), so the restatement calls the same primitives; ``fourier_unit_f64_dft`` is an independent
float64 numpy restatement (explicit DFT matrices, no FFT library) that pins the FourierUnit
semantics without torch.

Parity pinning: the reference ships NO tests / golden vectors for this path.
The oracle is pinned instead against outputs of the reference's own ``ffc.py`` classes executed in
the build container (``tests/golden/make_golden.py`` imports them from /root/reference with two
import stubs and commits the vectors under ``tests/golden/``); ``tests/test_oracle_golden.py``
replays them.
"""
from __future__ import annotations

import glob
import math
import os
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
BN_EPS = 1e-5  # nn.BatchNorm2d default, used by every BN on the path (ffc.py:60,131,236-239)

# ----------------------------------------------------------------------------------------------
# configuration (configs/training/big-lama.yaml:26-45, resolved)
# ----------------------------------------------------------------------------------------------

BIG_LAMA = dict(
    input_nc=4, output_nc=3, ngf=64, n_downsampling=3, n_blocks=18, add_out_act='sigmoid',
    init_conv_kwargs=dict(ratio_gin=0, ratio_gout=0, enable_lfu=False),
    downsample_conv_kwargs=dict(ratio_gin=0, ratio_gout=0, enable_lfu=False),
    resnet_conv_kwargs=dict(ratio_gin=0.75, ratio_gout=0.75, enable_lfu=False),
)


def small_config(ngf=8, n_blocks=2, n_downsampling=3, add_out_act='sigmoid'):
    """A structurally identical but tiny generator config for fast tests."""
    cfg = {k: (dict(v) if isinstance(v, dict) else v) for k, v in BIG_LAMA.items()}
    cfg.update(ngf=ngf, n_blocks=n_blocks, n_downsampling=n_downsampling, add_out_act=add_out_act)
    return cfg


def _split(channels: int, ratio: float) -> Tuple[int, int]:
    """(local, global) channel split, ffc.py:178-181 (int() truncation)."""
    cg = int(channels * ratio)
    return channels - cg, cg


def layer_plan(cfg: dict) -> List[dict]:
    """The nn.Sequential layout built by FFCResNetGenerator.__init__ (ffc.py:314-364).

    Returns one dict per top-level layer index, so ``plan[i]`` describes ``generator.model[i]``.
    """
    ngf, nd, nb = cfg['ngf'], cfg['n_downsampling'], cfg['n_blocks']
    maxf = cfg.get('max_features', 1024)
    init_kw, down_kw, res_kw = cfg['init_conv_kwargs'], cfg['downsample_conv_kwargs'], cfg['resnet_conv_kwargs']
    plan = [dict(kind='reflpad', pad=3),
            dict(kind='ffc_bn_act', cin=cfg['input_nc'], cout=ngf, k=7, stride=1, pad=0,
                 ratio_gin=init_kw['ratio_gin'], ratio_gout=init_kw['ratio_gout'])]
    for i in range(nd):
        mult = 2 ** i
        kw = dict(down_kw)
        if i == nd - 1:
            kw['ratio_gout'] = res_kw.get('ratio_gin', 0)  # ffc.py:322-324
        plan.append(dict(kind='ffc_bn_act', cin=min(maxf, ngf * mult), cout=min(maxf, ngf * mult * 2), k=3,
                         stride=2, pad=1, ratio_gin=kw['ratio_gin'], ratio_gout=kw['ratio_gout']))
    dim = min(maxf, ngf * 2 ** nd)
    for _ in range(nb):
        plan.append(dict(kind='resblock', dim=dim, ratio_gin=res_kw['ratio_gin'], ratio_gout=res_kw['ratio_gout']))
    plan.append(dict(kind='concat'))
    for i in range(nd):
        mult = 2 ** (nd - i)
        cin, cout = min(maxf, ngf * mult), min(maxf, int(ngf * mult / 2))
        plan += [dict(kind='convT', cin=cin, cout=cout), dict(kind='bn', c=cout), dict(kind='relu')]
    plan += [dict(kind='reflpad', pad=3), dict(kind='conv_out', cin=ngf, cout=cfg['output_nc'], k=7)]
    act = cfg.get('add_out_act', True)
    if act:
        plan.append(dict(kind='act', act='tanh' if act is True else act))  # ffc.py:362-363
    return plan


# ----------------------------------------------------------------------------------------------
# arithmetic (state-dict driven)
# ----------------------------------------------------------------------------------------------

def _bn(x: Tensor, sd: Dict[str, Tensor], p: str, calib: Optional[dict] = None) -> Tensor:
    """nn.BatchNorm2d.  Eval mode uses running stats; ``calib`` = one train-mode pass with momentum=1
    (running_mean <- batch mean, running_var <- unbiased batch var; normalisation by biased var)."""
    if calib is not None:
        dims = (0, 2, 3)
        mean = x.mean(dims)
        var_b = x.var(dims, unbiased=False)
        n = x.numel() // x.shape[1]
        sd[p + '.running_mean'] = mean.clone()
        sd[p + '.running_var'] = (var_b * (n / max(n - 1, 1))).clone()
        return (x - mean[None, :, None, None]) / torch.sqrt(var_b[None, :, None, None] + BN_EPS) \
            * sd[p + '.weight'][None, :, None, None] + sd[p + '.bias'][None, :, None, None]
    return F.batch_norm(x, sd[p + '.running_mean'], sd[p + '.running_var'], sd[p + '.weight'], sd[p + '.bias'],
                        training=False, eps=BN_EPS)


def _conv_reflect(x: Tensor, w: Tensor, stride: int, pad: int) -> Tensor:
    """nn.Conv2d(padding_mode='reflect', bias=False) = F.pad(mode='reflect') + valid conv (ffc.py:188-196)."""
    if pad > 0:
        x = F.pad(x, (pad, pad, pad, pad), mode='reflect')
    return F.conv2d(x, w, None, stride=stride)


def fourier_unit(x: Tensor, sd: Dict[str, Tensor], p: str, calib=None) -> Tensor:
    """FourierUnit.forward, ffc.py:76-113 with every optional branch off (as in all shipped configs)."""
    b, c, h, w = x.shape
    ff = torch.fft.rfftn(x, dim=(-2, -1), norm='ortho')                       # ffc.py:86
    ff = torch.stack((ff.real, ff.imag), dim=-1)                               # ffc.py:87
    ff = ff.permute(0, 1, 4, 2, 3).contiguous().view(b, -1, h, w // 2 + 1)    # ffc.py:88-89 (2c=Re, 2c+1=Im)
    ff = F.conv2d(ff, sd[p + '.conv_layer.weight'])                            # ffc.py:100
    ff = torch.relu(_bn(ff, sd, p + '.bn', calib))                             # ffc.py:101
    ff = ff.view(b, -1, 2, h, w // 2 + 1).permute(0, 1, 3, 4, 2).contiguous()  # ffc.py:103-104
    ff = torch.complex(ff[..., 0], ff[..., 1])                                 # ffc.py:105
    return torch.fft.irfftn(ff, s=(h, w), dim=(-2, -1), norm='ortho')          # ffc.py:107-108


def spectral_transform(x: Tensor, sd, p: str, calib=None) -> Tensor:
    """SpectralTransform.forward, ffc.py:142-163 (stride 1 -> Identity downsample, LFU off -> xs = 0)."""
    x = F.conv2d(x, sd[p + '.conv1.0.weight'])                                 # ffc.py:145
    x = torch.relu(_bn(x, sd, p + '.conv1.1', calib))
    out = fourier_unit(x, sd, p + '.fu', calib)                                # ffc.py:146
    return F.conv2d(x + out, sd[p + '.conv2.weight'])                          # ffc.py:161


def ffc(x_l: Tensor, x_g, sd, p: str, spec: dict, calib=None):
    """FFC.forward, ffc.py:205-225 (gated=False)."""
    k, stride, pad = spec['k'], spec['stride'], spec['pad']
    out_l, out_g = 0, 0
    has_gin = torch.is_tensor(x_g)
    if spec['ratio_gout'] != 1:                                                # ffc.py:220-221
        out_l = _conv_reflect(x_l, sd[p + '.convl2l.weight'], stride, pad)
        if has_gin:
            out_l = out_l + _conv_reflect(x_g, sd[p + '.convg2l.weight'], stride, pad)
    if spec['ratio_gout'] != 0:                                                # ffc.py:222-223
        out_g = _conv_reflect(x_l, sd[p + '.convl2g.weight'], stride, pad)
        if has_gin:
            out_g = out_g + spectral_transform(x_g, sd, p + '.convg2g', calib)
    return out_l, out_g


def ffc_bn_act(x_l, x_g, sd, p: str, spec: dict, calib=None):
    """FFC_BN_ACT.forward, ffc.py:251-255; activation is ReLU for every layer of the generator."""
    out_l, out_g = ffc(x_l, x_g, sd, p + '.ffc', spec, calib)
    if torch.is_tensor(out_l):
        out_l = torch.relu(_bn(out_l, sd, p + '.bn_l', calib))
    if torch.is_tensor(out_g):
        out_g = torch.relu(_bn(out_g, sd, p + '.bn_g', calib))
    return out_l, out_g


def ffc_resnet_block(x_l, x_g, sd, p: str, spec: dict, calib=None):
    """FFCResnetBlock.forward, ffc.py:277-292 (inline=False): two FFC_BN_ACT then the residual add."""
    s = dict(k=3, stride=1, pad=1, ratio_gin=spec['ratio_gin'], ratio_gout=spec['ratio_gout'])
    y_l, y_g = ffc_bn_act(x_l, x_g, sd, p + '.conv1', s, calib)
    y_l, y_g = ffc_bn_act(y_l, y_g, sd, p + '.conv2', s, calib)
    return x_l + y_l, x_g + y_g                                                # ffc.py:288


def run_layers(x, sd, cfg: dict, start: int = 0, stop: Optional[int] = None, prefix: str = 'model.',
               calib=None, taps: Optional[dict] = None):
    """Run ``generator.model[start:stop]`` (ffc.py:366-367).  ``x`` is a tensor or the (x_l, x_g) tuple.
    ``taps``: optional dict filled with the output of every layer index (for per-layer goldens)."""
    plan = layer_plan(cfg)
    stop = len(plan) if stop is None else stop
    for i in range(start, stop):
        L, p = plan[i], f'{prefix}{i}'
        kind = L['kind']
        if kind == 'reflpad':
            x = F.pad(x, (L['pad'],) * 4, mode='reflect')
        elif kind == 'ffc_bn_act':
            x_l, x_g = x if isinstance(x, tuple) else (x, 0)
            x = ffc_bn_act(x_l, x_g, sd, p, L, calib)
        elif kind == 'resblock':
            x = ffc_resnet_block(x[0], x[1], sd, p, L, calib)
        elif kind == 'concat':                                                 # ffc.py:296-302
            x = torch.cat(x, dim=1) if torch.is_tensor(x[1]) else x[0]
        elif kind == 'convT':                                                  # ffc.py:348-351
            x = F.conv_transpose2d(x, sd[p + '.weight'], sd[p + '.bias'], stride=2, padding=1, output_padding=1)
        elif kind == 'bn':
            x = _bn(x, sd, p, calib)
        elif kind == 'relu':
            x = torch.relu(x)
        elif kind == 'conv_out':                                               # ffc.py:360-361
            x = F.conv2d(x, sd[p + '.weight'], sd[p + '.bias'])
        elif kind == 'act':
            x = torch.sigmoid(x) if L['act'] == 'sigmoid' else torch.tanh(x)
        else:
            raise ValueError(kind)
        if taps is not None:
            taps[i] = x
    return x


def generator_forward(x: Tensor, sd, cfg: dict, prefix: str = 'model.', calib=None, taps=None) -> Tensor:
    """FFCResNetGenerator.forward, ffc.py:366-367."""
    return run_layers(x, sd, cfg, 0, None, prefix, calib, taps)


class _HalfOperands:
    """``torch.nn.functional`` with the operands of every conv rounded to fp16 (products exact, fp32 accumulation);
    ``weights=False``: only the activations are rounded (LAMA_PREC_F16 since round 3: the weights keep their hi + lo parts)."""

    def __init__(self, weights: bool = True):
        self.weights = weights

    def __getattr__(self, name):
        return getattr(torch.nn.functional, name)

    def conv2d(self, x, w, b=None, **kw):
        return torch.nn.functional.conv2d(x.half().float(), w.half().float() if self.weights else w, b, **kw)

    def conv_transpose2d(self, x, w, b=None, **kw):
        return torch.nn.functional.conv_transpose2d(x.half().float(), w.half().float() if self.weights else w, b, **kw)


def generator_forward_fp16_emulated(x: Tensor, sd, cfg: dict, prefix: str = 'model.', weights: bool = True) -> Tensor:
    """What a half-precision run of the same network costs (BASELINE configs[2]; the reference never runs FFC in fp16,
    configs/training/trainer/any_gpu_large_ssim_ddp_final.yaml:13-16 keeps ``precision: 16`` commented out): every conv / GEMM of
    ``generator_forward`` with BOTH operands rounded to fp16, exact products, fp32 accumulation, everything else fp32.  This is
    the yardstick of the LAMA_PREC_F16 tests: the HIP path (fp16 tensors in HBM, BatchNorm folded before the rounding) is asked to
    stay at this error level against the fp32 oracle, not at the fp32 paths' 2e-4.  ``weights=False``: only the conv INPUTS are
    rounded -- the yardstick of the two-product form (hi + lo weights) the HIP path uses since round 3."""
    global F
    keep = F
    F = _HalfOperands(weights)
    try:
        return generator_forward(x, sd, cfg, prefix)
    finally:
        F = keep


FP16_GROUPS = ('front', 'blockin', 'mid', 'x1', 'spec', 't', 'up1', 'up2', 'up3')


def generator_forward_fp16_storage(x: Tensor, sd, cfg: dict, half=('front', 'blockin', 'mid', 'x1', 'spec', 't'), prefix: str = 'model.') -> Tensor:
    """``generator_forward`` with a rounding to fp16 exactly where a LAMA_PREC_F16 run STORES an fp16 tensor in HBM (weights and all arithmetic
    fp32: the path keeps hi + lo weight parts and accumulates in fp32) -- the yardstick of the fp16-activation path by tensor group:
    ``front`` stem / down1 / down2 outputs; ``mid`` the (x_l | x_g) tensor between conv1 and conv2 of a resnet block (the residual stream is
    fp32); ``x1`` SpectralTransform.conv1's output; ``spec`` both spectra of the FourierUnit; ``t`` = x1 + fu(x1); ``up1..3`` the outputs of the
    three ConvTranspose2d + BN + ReLU; and ``blockin``: not a stored tensor but the READ of the fp32 residual stream by conv1 of every resnet
    block -- LAMA_PREC_F16 spends one matrix-core operand on an activation, so the stream is rounded to fp16 while it is staged (the stream
    itself, and the residual add, stay fp32).  Default = the layout since round 4 (the tail behind the blocks stays fp32 with the 3-term
    split: FFCResNetGenerator.f16_fp32_tail)."""
    half = set(half)

    def r(t, group):
        return t.half().float() if group in half else t

    def fu(x, p):
        b, c, h, w = x.shape
        ff = torch.fft.rfftn(x, dim=(-2, -1), norm='ortho')
        ff = torch.stack((ff.real, ff.imag), dim=-1).permute(0, 1, 4, 2, 3).contiguous().view(b, -1, h, w // 2 + 1)
        ff = F.conv2d(r(ff, 'spec'), sd[p + '.conv_layer.weight'])
        ff = r(torch.relu(_bn(ff, sd, p + '.bn')), 'spec')
        ff = ff.view(b, -1, 2, h, w // 2 + 1).permute(0, 1, 3, 4, 2).contiguous()
        return torch.fft.irfftn(torch.complex(ff[..., 0], ff[..., 1]), s=(h, w), dim=(-2, -1), norm='ortho')

    def st(x, p, calib=None):
        x = r(torch.relu(_bn(F.conv2d(x, sd[p + '.conv1.0.weight']), sd, p + '.conv1.1')), 'x1')
        return F.conv2d(r(x + fu(x, p + '.fu'), 't'), sd[p + '.conv2.weight'])

    global spectral_transform
    keep = spectral_transform
    spectral_transform = lambda x, sd_, p, calib=None: st(x, p)
    try:
        plan, n_up = layer_plan(cfg), 0
        for i, L in enumerate(plan):
            p, kind = f'{prefix}{i}', L['kind']
            if kind == 'reflpad':
                x = F.pad(x, (L['pad'],) * 4, mode='reflect')
            elif kind == 'ffc_bn_act':
                x_l, x_g = x if isinstance(x, tuple) else (x, 0)
                x = ffc_bn_act(x_l, x_g, sd, p, L)
                if not torch.is_tensor(x[1]):                 # stem / down1 / down2: local only
                    x = (r(x[0], 'front'), x[1])
            elif kind == 'resblock':
                s = dict(k=3, stride=1, pad=1, ratio_gin=L['ratio_gin'], ratio_gout=L['ratio_gout'])
                y_l, y_g = ffc_bn_act(r(x[0], 'blockin'), r(x[1], 'blockin'), sd, p + '.conv1', s)
                y_l, y_g = ffc_bn_act(r(y_l, 'mid'), r(y_g, 'mid'), sd, p + '.conv2', s)
                x = (x[0] + y_l, x[1] + y_g)
            elif kind == 'concat':
                x = torch.cat(x, dim=1) if torch.is_tensor(x[1]) else x[0]
            elif kind == 'convT':
                x = F.conv_transpose2d(x, sd[p + '.weight'], sd[p + '.bias'], stride=2, padding=1, output_padding=1)
            elif kind == 'bn':
                x = _bn(x, sd, p)
            elif kind == 'relu':
                n_up += 1
                x = r(torch.relu(x), f'up{n_up}')
            elif kind == 'conv_out':
                x = F.conv2d(x, sd[p + '.weight'], sd[p + '.bias'])
            elif kind == 'act':
                x = torch.sigmoid(x) if L['act'] == 'sigmoid' else torch.tanh(x)
            else:
                raise ValueError(kind)
        return x
    finally:
        spectral_transform = keep


def training_module_forward(batch: dict, sd, cfg: dict, prefix: str = 'generator.model.') -> dict:
    """DefaultInpaintingTrainingModule.forward, eval path: trainers/default.py:56-59,67-71,82-86."""
    img, mask = batch['image'], batch['mask']
    masked_img = img * (1 - mask)                                              # default.py:59
    masked_img = torch.cat([masked_img, mask], dim=1)                          # default.py:67-68 (concat_mask)
    batch['predicted_image'] = generator_forward(masked_img.float(), sd, cfg, prefix)   # default.py:70
    batch['inpainted'] = mask * batch['predicted_image'] + (1 - mask) * batch['image']  # default.py:71
    batch['mask_for_losses'] = mask                                            # default.py:82-84
    return batch


# ----------------------------------------------------------------------------------------------
# independent float64 restatement of FourierUnit (numpy, explicit DFT matrices)
# ----------------------------------------------------------------------------------------------

def fourier_unit_f64_dft(x: np.ndarray, w: np.ndarray, gamma, beta, mean, var) -> np.ndarray:
    """FourierUnit (ffc.py:76-113) with no FFT library.  x [B,C,h,w]; w [2Co,2Ci] (1x1 conv matrix).

    Pins (a) the Re/Im interleave ``2c -> Re, 2c+1 -> Im`` on both sides of the conv (ffc.py:87-89,
    103-105), (b) BN(eval)+ReLU per real channel, (c) the irfftn semantics on a non-Hermitian
    spectrum: complex inverse DFT along h, then c2r along w that ignores Im of bins 0 and w/2
    (SURVEY.md K6).
    """
    x = np.asarray(x, np.float64)
    b, c, h, wd = x.shape
    wf = wd // 2 + 1
    # forward: ortho 2-D DFT, keep the first wf columns
    fh = np.exp(-2j * np.pi * np.outer(np.arange(h), np.arange(h)) / h)
    fw = np.exp(-2j * np.pi * np.outer(np.arange(wd), np.arange(wf)) / wd)
    spec = np.einsum('uy,bcyx,xk->bcuk', fh, x, fw) / math.sqrt(h * wd)
    st = np.stack([spec.real, spec.imag], axis=2).reshape(b, 2 * c, h, wf)
    y = np.einsum('oi,biuk->bouk', np.asarray(w, np.float64), st)
    scale = np.asarray(gamma, np.float64) / np.sqrt(np.asarray(var, np.float64) + BN_EPS)
    shift = np.asarray(beta, np.float64) - np.asarray(mean, np.float64) * scale
    y = np.maximum(y * scale[None, :, None, None] + shift[None, :, None, None], 0.0)
    co = y.shape[1] // 2
    y = y.reshape(b, co, 2, h, wf)
    yc = y[:, :, 0] + 1j * y[:, :, 1]
    # inverse: complex IDFT along h ...
    ih = np.exp(+2j * np.pi * np.outer(np.arange(h), np.arange(h)) / h) / math.sqrt(h)
    z = np.einsum('yu,bcuk->bcyk', ih, yc)
    # ... then c2r along w: out[x] = (1/sqrt w) [Re z0 + (-1)^x Re z_{w/2} + 2 sum_k Re(z_k e^{2 pi i k x / w})]
    xs = np.arange(wd)
    out = np.repeat(z[..., 0:1].real, wd, axis=-1).copy()
    last = wf - 1
    if wd % 2 == 0:
        out += z[..., last:last + 1].real * ((-1.0) ** xs)
        mid = range(1, last)
    else:
        mid = range(1, last + 1)
    for k in mid:
        ph = np.exp(2j * np.pi * k * xs / wd)
        out += 2.0 * (z[..., k:k + 1] * ph).real
    return out / math.sqrt(wd)


# ----------------------------------------------------------------------------------------------
# synthetic fixtures (no checkpoint exists offline: SURVEY.md section 8c)
# ----------------------------------------------------------------------------------------------

def state_dict_spec(cfg: dict, prefix: str = 'model.') -> List[Tuple[str, Tuple[int, ...], str]]:
    """(key, shape, role) for every tensor of the reference generator state_dict.
    role in {'conv', 'convT', 'bias', 'bn'}; a 'bn' entry expands to weight/bias/running_mean/
    running_var/num_batches_tracked."""
    out = []
    for i, L in enumerate(layer_plan(cfg)):
        p = f'{prefix}{i}'
        if L['kind'] == 'ffc_bn_act':
            out += _ffc_bn_act_spec(p, L)
        elif L['kind'] == 'resblock':
            s = dict(cin=L['dim'], cout=L['dim'], k=3, ratio_gin=L['ratio_gin'], ratio_gout=L['ratio_gout'])
            out += _ffc_bn_act_spec(p + '.conv1', s) + _ffc_bn_act_spec(p + '.conv2', s)
        elif L['kind'] == 'convT':
            out += [(p + '.weight', (L['cin'], L['cout'], 3, 3), 'convT'), (p + '.bias', (L['cout'],), 'bias')]
        elif L['kind'] == 'bn':
            out.append((p, (L['c'],), 'bn'))
        elif L['kind'] == 'conv_out':
            out += [(p + '.weight', (L['cout'], L['cin'], L['k'], L['k']), 'conv'), (p + '.bias', (L['cout'],), 'bias')]
    return out


def _ffc_bn_act_spec(p: str, L: dict):
    in_cl, in_cg = _split(L['cin'], L['ratio_gin'])
    out_cl, out_cg = _split(L['cout'], L['ratio_gout'])
    k = L['k']
    out = []
    if in_cl and out_cl:
        out.append((p + '.ffc.convl2l.weight', (out_cl, in_cl, k, k), 'conv'))
    if in_cl and out_cg:
        out.append((p + '.ffc.convl2g.weight', (out_cg, in_cl, k, k), 'conv'))
    if in_cg and out_cl:
        out.append((p + '.ffc.convg2l.weight', (out_cl, in_cg, k, k), 'conv'))
    if in_cg and out_cg:
        q, half = p + '.ffc.convg2g', out_cg // 2
        out += [(q + '.conv1.0.weight', (half, in_cg, 1, 1), 'conv'), (q + '.conv1.1', (half,), 'bn'),
                (q + '.fu.conv_layer.weight', (2 * half, 2 * half, 1, 1), 'conv'), (q + '.fu.bn', (2 * half,), 'bn'),
                (q + '.conv2.weight', (out_cg, half, 1, 1), 'conv')]
    if out_cl:
        out.append((p + '.bn_l', (out_cl,), 'bn'))
    if out_cg:
        out.append((p + '.bn_g', (out_cg,), 'bn'))
    return out


def make_synthetic_state_dict(cfg: dict, seed: int = 0, calib_hw: int = 64, prefix: str = 'model.') -> Dict[str, Tensor]:
    """Seeded big-lama-shaped weights with calibrated BatchNorm statistics.

    conv / convT weights ~ U(-1/sqrt(fan_in), 1/sqrt(fan_in)) (PyTorch's default init distribution),
    BN gamma ~ U(0.5, 1.5), beta ~ 0.2 N(0,1); running stats come from one calibration pass (train-mode
    BN semantics, momentum 1) over a seeded synthetic batch, so that activations stay O(1) through
    all blocks and the sigmoid output spans (0,1) (un-calibrated default init collapses to 0.5).
    """
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, Tensor] = {}
    for key, shape, role in state_dict_spec(cfg, prefix):
        if role in ('conv', 'convT'):
            fan_in = shape[1] * shape[2] * shape[3] if role == 'conv' else shape[0] * shape[2] * shape[3]
            bound = 1.0 / math.sqrt(fan_in)
            sd[key] = (torch.rand(shape, generator=g) * 2 - 1) * bound
        elif role == 'bias':
            sd[key] = (torch.rand(shape, generator=g) * 2 - 1) * 0.1
        else:
            sd[key + '.weight'] = torch.rand(shape, generator=g) + 0.5
            sd[key + '.bias'] = 0.2 * torch.randn(shape, generator=g)
            sd[key + '.running_mean'] = torch.zeros(shape)
            sd[key + '.running_var'] = torch.ones(shape)
            sd[key + '.num_batches_tracked'] = torch.tensor(1, dtype=torch.int64)
    batch = make_synthetic_batch(2, calib_hw, calib_hw, seed=seed + 1000)
    x = torch.cat([batch['image'] * (1 - batch['mask']), batch['mask']], 1)
    if cfg['input_nc'] != 4:
        x = torch.rand(2, cfg['input_nc'], calib_hw, calib_hw, generator=g)
    with torch.no_grad():
        generator_forward(x, sd, cfg, prefix, calib={})
    return sd


def make_synthetic_batch(b: int, h: int, w: int, seed: int = 1234) -> Dict[str, Tensor]:
    """image [b,3,h,w] uniform [0,1) quantised to u8/255; mask [b,1,h,w] in {0,1}: a centred rectangle
    (25% of the area) plus three random strokes."""
    g = torch.Generator().manual_seed(seed)
    img = torch.floor(torch.rand(b, 3, h, w, generator=g) * 256).clamp_(0, 255) / 255.0
    mask = torch.zeros(b, 1, h, w)
    mask[:, :, h // 4: h // 4 + h // 2, w // 4: w // 4 + w // 2] = 1.0
    for bi in range(b):
        for _ in range(3):
            y0, x0 = int(torch.randint(0, h, (1,), generator=g)), int(torch.randint(0, w, (1,), generator=g))
            ln = int(torch.randint(max(2, w // 8), max(3, w // 2), (1,), generator=g))
            th = max(1, h // 32)
            if int(torch.randint(0, 2, (1,), generator=g)):
                mask[bi, 0, y0:y0 + th, x0:x0 + ln] = 1.0
            else:
                mask[bi, 0, y0:y0 + ln, x0:x0 + th] = 1.0
    return dict(image=img, mask=mask)


# ----------------------------------------------------------------------------------------------
# predict.py glue (restated with PIL/numpy; cv2 / hydra are not installed)
# ----------------------------------------------------------------------------------------------

def load_image(fname: str, mode: str = 'RGB') -> np.ndarray:
    """saicinpainting/evaluation/data.py:12-20."""
    from PIL import Image
    img = np.array(Image.open(fname).convert(mode))
    if img.ndim == 3:
        img = np.transpose(img, (2, 0, 1))
    return img.astype('float32') / 255


def ceil_modulo(x: int, mod: int) -> int:
    """saicinpainting/evaluation/data.py:23-26."""
    return x if x % mod == 0 else (x // mod + 1) * mod


def pad_img_to_modulo(img: np.ndarray, mod: int) -> np.ndarray:
    """saicinpainting/evaluation/data.py:29-33 (np.pad mode='symmetric', bottom/right only)."""
    c, h, w = img.shape
    return np.pad(img, ((0, 0), (0, ceil_modulo(h, mod) - h), (0, ceil_modulo(w, mod) - w)), mode='symmetric')


def list_dataset(indir: str, img_suffix: str = '.png') -> List[Tuple[str, str]]:
    """InpaintingDataset.__init__, evaluation/data.py:59-62: (mask path, image path) pairs."""
    masks = sorted(glob.glob(os.path.join(indir, '**', '*mask*.png'), recursive=True))
    return [(m, m.rsplit('_mask', 1)[0] + img_suffix) for m in masks]


def predict_one(image: np.ndarray, mask: np.ndarray, sd, cfg: dict, pad_mod: int = 8,
                prefix: str = 'generator.model.') -> Tuple[np.ndarray, np.ndarray]:
    """One iteration of the bin/predict.py:67-94 loop for decoded ``image`` [3,H,W] and ``mask`` [H,W]
    (floats in [0,1]).  Returns (float inpainted [H,W,3] cropped to the unpadded size, uint8 RGB image)."""
    h, w = image.shape[1:]
    img_p = pad_img_to_modulo(image, pad_mod)                                  # data.py:78-81
    msk_p = pad_img_to_modulo(mask[None, ...], pad_mod)
    batch = dict(image=torch.from_numpy(img_p)[None], mask=torch.from_numpy(msk_p)[None])
    batch['mask'] = (batch['mask'] > 0) * 1                                    # predict.py:84 (int64)
    with torch.no_grad():
        batch = training_module_forward(batch, sd, cfg, prefix)
    cur = batch['inpainted'][0].permute(1, 2, 0).numpy()[:h, :w]               # predict.py:86-90
    u8 = np.clip(cur * 255, 0, 255).astype('uint8')                            # predict.py:92 (truncation)
    return cur, u8
