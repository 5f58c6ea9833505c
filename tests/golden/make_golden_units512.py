#!/usr/bin/env python3
"""Golden vectors of the FFC units at BIG-LAMA CHANNEL COUNTS from the reference's own classes (Next #6).

Runs only in the build container (needs /root/reference).  The reference ``FFCResnetBlock(512, ratio 0.75)`` -- and, inside it, its
``FFC_BN_ACT``, ``SpectralTransform`` and ``FourierUnit`` (ffc.py:49-292) -- is instantiated UNMODIFIED (the two import stubs of
make_golden.py), loaded with block ``model.5`` of the seeded synthetic big-lama state dict (seed 0, the one biglama_256.npz uses; it is
regenerated from the seed at test time) and run on a seeded [2, 512, 64, 64] state at the bottleneck shape of BASELINE configs[1].

  ffc_block512.npz   strided samples ([:, :, 1::8, ::4]) + (mean, std, absmax) of
                     block output (x_l, x_g), conv1 layer output (x_l, x_g), SpectralTransform output, FourierUnit output
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

BLOCK = 'model.5'


def block_inputs(seed=21, batch=2, hw=64):
    """The seeded (x_l | x_g) state the units are run on: non-negative local / global features of O(1), as behind the downsampling
    layers (ReLU outputs), with one plane of larger values so that the spectral branch sees a strong DC bin."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(batch, 512, hw, hw, generator=g).abs()
    x[:, 130] *= 4.0
    return x[:, :128].contiguous(), x[:, 128:].contiguous()


def sample(t):
    return t[:, :, 1::8, ::4].contiguous().numpy()


def stat(t):
    t = t.double()
    return np.array([t.mean().item(), t.std().item(), t.abs().max().item()])


def main():
    import torch.nn as nn
    from make_golden import import_reference
    from oracle import lama_oracle as O
    ref_ffc, _ = import_reference()
    sd = O.make_synthetic_state_dict(O.BIG_LAMA, seed=0, calib_hw=64)
    bsd = {k[len(BLOCK) + 1:]: v for k, v in sd.items() if k.startswith(BLOCK + '.')}
    blk = ref_ffc.FFCResnetBlock(512, padding_type='reflect', norm_layer=nn.BatchNorm2d, activation_layer=nn.ReLU,
                                 **O.BIG_LAMA['resnet_conv_kwargs']).eval()
    res = blk.load_state_dict(bsd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    xl, xg = block_inputs()
    with torch.no_grad():
        yl, yg = blk((xl, xg))
        l1, g1 = blk.conv1((xl, xg))
        st = blk.conv1.ffc.convg2g(xg)
        fu = blk.conv1.ffc.convg2g.fu(xg[:, :192].contiguous())          # the FourierUnit alone, on the first 192 global channels
    out = dict(sd_checksum=np.array([sum(float(v.double().sum()) for v in bsd.values() if v.is_floating_point())]),
               x_checksum=np.array([float(xl.double().sum()), float(xg.double().sum())]))
    for name, t in dict(yl=yl, yg=yg, c1_l=l1, c1_g=g1, st=st, fu=fu).items():
        out[name + '_sample'] = sample(t)
        out[name + '_stat'] = stat(t)
    np.savez_compressed(os.path.join(HERE, 'ffc_block512.npz'), **out)
    print('written', os.path.join(HERE, 'ffc_block512.npz'), {k: v.shape for k, v in out.items()})


if __name__ == '__main__':
    main()
