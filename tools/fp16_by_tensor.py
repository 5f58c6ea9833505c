#!/usr/bin/env python3
"""CPU experiment (ORACLE only, test infrastructure): which STORED tensors of the LAMA_PREC_F16 path (BASELINE configs[2]: fp16 activations in HBM)
carry its end-to-end error (Next #7).  The fp32 oracle is run with roundings to fp16 inserted exactly where the HIP path stores an fp16
tensor (weights stay fp32: the path keeps hi + lo weight parts); every group below can be switched to fp32 storage:

    front   stem / down1 / down2 outputs (the input of down3 .. is read as fp16)
    mid     the (x_l | x_g) tensor between conv1 and conv2 of every FFCResnetBlock            (the residual stream itself is fp32 already)
    x1      SpectralTransform.conv1's output (input of the FourierUnit and of the residual add x1 + fu(x1))
    spec    the two spectra of the FourierUnit (rfft2 output / spectral 1x1 output)
    t       x1 + fu(x1), the input of SpectralTransform.conv2
    up      the outputs of the three ConvTranspose2d + BN + ReLU (up3 = the head's input); up1 / up2 / up3 select one of them

    python tools/fp16_by_tensor.py [res=1024] [batch=1]
"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import lama_oracle as O
torch.set_num_threads(int(os.environ.get('OMP_NUM_THREADS', '8')))
Fn = torch.nn.functional
HALF = set()


def r(t, group):
    return t.half().float() if group in HALF else t


def fourier_unit(x, sd, p, calib=None):
    b, c, h, w = x.shape
    ff = torch.fft.rfftn(x, dim=(-2, -1), norm='ortho')
    ff = torch.stack((ff.real, ff.imag), dim=-1).permute(0, 1, 4, 2, 3).contiguous().view(b, -1, h, w // 2 + 1)
    ff = r(ff, 'spec')
    ff = Fn.conv2d(ff, sd[p + '.conv_layer.weight'])
    ff = r(torch.relu(O._bn(ff, sd, p + '.bn', calib)), 'spec')
    ff = ff.view(b, -1, 2, h, w // 2 + 1).permute(0, 1, 3, 4, 2).contiguous()
    return torch.fft.irfftn(torch.complex(ff[..., 0], ff[..., 1]), s=(h, w), dim=(-2, -1), norm='ortho')


def spectral_transform(x, sd, p, calib=None):
    x = Fn.conv2d(x, sd[p + '.conv1.0.weight'])
    x = r(torch.relu(O._bn(x, sd, p + '.conv1.1', calib)), 'x1')
    out = fourier_unit(x, sd, p + '.fu', calib)
    return Fn.conv2d(r(x + out, 't'), sd[p + '.conv2.weight'])


def ffc_resnet_block(x_l, x_g, sd, p, spec, calib=None):
    s = dict(k=3, stride=1, pad=1, ratio_gin=spec['ratio_gin'], ratio_gout=spec['ratio_gout'])
    y_l, y_g = O.ffc_bn_act(x_l, x_g, sd, p + '.conv1', s, calib)
    y_l, y_g = O.ffc_bn_act(r(y_l, 'mid'), r(y_g, 'mid'), sd, p + '.conv2', s, calib)
    return x_l + y_l, x_g + y_g


def run(x, sd, cfg):
    plan = O.layer_plan(cfg)
    n_up = 0
    for i, L in enumerate(plan):
        p, kind = f'model.{i}', L['kind']
        if kind == 'reflpad':
            x = Fn.pad(x, (L['pad'],) * 4, mode='reflect')
        elif kind == 'ffc_bn_act':
            x_l, x_g = x if isinstance(x, tuple) else (x, 0)
            x = O.ffc_bn_act(x_l, x_g, sd, p, L, None)
            if not torch.is_tensor(x[1]):                     # stem / down1 / down2: local only -> an fp16 tensor in HBM
                x = (r(x[0], 'front'), x[1])
        elif kind == 'resblock':
            x = ffc_resnet_block(x[0], x[1], sd, p, L)
        elif kind == 'concat':
            x = torch.cat(x, dim=1) if torch.is_tensor(x[1]) else x[0]
        elif kind == 'convT':
            x = Fn.conv_transpose2d(x, sd[p + '.weight'], sd[p + '.bias'], stride=2, padding=1, output_padding=1)
        elif kind == 'bn':
            x = O._bn(x, sd, p, None)
        elif kind == 'relu':
            x = torch.relu(x)
            n_up += 1
            x = r(r(x, 'up'), f'up{n_up}')
        elif kind == 'conv_out':
            x = Fn.conv2d(x, sd[p + '.weight'], sd[p + '.bias'])
        elif kind == 'act':
            x = torch.sigmoid(x) if L['act'] == 'sigmoid' else torch.tanh(x)
    return x


def main():
    res = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    bn = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    cfg = O.BIG_LAMA
    sd = O.make_synthetic_state_dict(cfg, seed=0, calib_hw=64)
    batch = O.make_synthetic_batch(bn, res, res, seed=12)
    x = torch.cat([batch['image'] * (1 - batch['mask']), batch['mask']], 1)
    keep = O.spectral_transform
    O.spectral_transform = spectral_transform
    ALL = ['front', 'mid', 'x1', 'spec', 't', 'up']
    try:
        with torch.no_grad():
            HALF.clear()
            ref = run(x, sd, cfg)
            assert float((ref - O.generator_forward(x, sd, cfg)).abs().max()) == 0.0
            cases = [('all fp16 (the shipped LAMA_PREC_F16 layout)', ALL)] + [(f'only {g}', [g]) for g in ALL] + \
                    [('only up1', ['up1']), ('only up2', ['up2']), ('only up3', ['up3']),
                     ('all but up3 (head input fp32)', ['front', 'mid', 'x1', 'spec', 't', 'up1', 'up2']),
                     ('all but up2, up3', ['front', 'mid', 'x1', 'spec', 't', 'up1']),
                     ('all but x1, t, spec', ['front', 'mid', 'up']), ('all but spec', ['front', 'mid', 'x1', 't', 'up']),
                     ('all but mid', ['front', 'x1', 'spec', 't', 'up']), ('front + up only', ['front', 'up'])]
            for name, groups in cases:
                HALF.clear(); HALF.update(groups)
                d = (run(x, sd, cfg) - ref).abs()
                print(f'{bn} x {res}^2  fp16 storage: {name:45s} max-abs {float(d.max()):.2e}  mean-abs {float(d.mean()):.2e}', flush=True)
    finally:
        O.spectral_transform = keep


if __name__ == '__main__':
    main()
