#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REFERENCE's own classes.

Runs only in the build container (needs /root/reference).  It imports the reference
``saicinpainting/training/modules/ffc.py`` unmodified, with two import stubs for packages that are
not installed and never executed on the big-lama path (kornia's ``rotate`` and
pytorch_lightning's ``seed_everything``), instantiates ``FFCResNetGenerator`` through the reference
factory ``make_generator(kind='ffc_resnet')``, loads the seeded synthetic state dict produced by
``oracle.lama_oracle.make_synthetic_state_dict`` with ``strict=True`` (which also pins the key map),
and records reference outputs:

  small_gen.npz      full tensors for a tiny generator (ngf=8, 2 blocks): weights are regenerated
                     from the seed at test time; inputs, per-layer taps and the output are stored.
  ffc_units.npz      FourierUnit / SpectralTransform / FFC_BN_ACT / FFCResnetBlock in isolation at
                     odd and even sizes (pins irfftn on a non-Hermitian spectrum).
  biglama_256.npz    big-lama shape (50 975 875 params; weights regenerated from seed 0): a strided
                     sample of the output and per-layer (mean, std, absmax) for 1x4x256x256.
  predict_glue.npz   decode/pad/blend/u8 glue (DefaultInpaintingTrainingModule.forward arithmetic).

The vectors are small (< 2 MB total) and committed; this script is committed beside them.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, ROOT)


def import_reference():
    """Import the reference modules with the two stubs."""
    if 'kornia' not in sys.modules:
        k = types.ModuleType('kornia'); kg = types.ModuleType('kornia.geometry')
        kt = types.ModuleType('kornia.geometry.transform')
        kt.rotate = lambda *a, **kw: (_ for _ in ()).throw(RuntimeError('stub'))
        k.geometry = kg; kg.transform = kt
        sys.modules.update({'kornia': k, 'kornia.geometry': kg, 'kornia.geometry.transform': kt})
    if 'pytorch_lightning' not in sys.modules:
        pl = types.ModuleType('pytorch_lightning')
        pl.seed_everything = lambda *a, **kw: None
        sys.modules['pytorch_lightning'] = pl
    if REF not in sys.path:
        sys.path.insert(0, REF)
    from saicinpainting.training.modules import ffc as ref_ffc
    from saicinpainting.training.modules import make_generator
    return ref_ffc, make_generator


def ref_generator(cfg, sd):
    ref_ffc, make_generator = import_reference()
    gen = make_generator(None, kind='ffc_resnet', **cfg)
    missing = gen.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return gen.eval()


def taps_of(gen, x):
    taps = {}
    with torch.no_grad():
        for i, layer in enumerate(gen.model):
            x = layer(x)
            taps[i] = x
    return x, taps


def stat(t):
    t = t.double()
    return np.array([t.mean().item(), t.std().item(), t.abs().max().item()])


def main():
    from oracle import lama_oracle as O
    torch.manual_seed(0)
    out = {}

    # ---- small generator, full tensors -----------------------------------------------------
    cfg = O.small_config(ngf=8, n_blocks=2)
    sd = O.make_synthetic_state_dict(cfg, seed=7, calib_hw=32)
    gen = ref_generator(cfg, sd)
    small = {}
    for name, (b, h, w) in dict(a=(2, 32, 32), b=(1, 40, 56), c=(1, 24, 72)).items():
        batch = O.make_synthetic_batch(b, h, w, seed=11)
        x = torch.cat([batch['image'] * (1 - batch['mask']), batch['mask']], 1)
        y, taps = taps_of(gen, x)
        small[f'{name}_x'] = x.numpy()
        small[f'{name}_y'] = y.numpy()
        for i in (4, 5, 6, 7, 10, 13):
            t = taps[i]
            if isinstance(t, tuple):
                small[f'{name}_tap{i}_l'] = t[0].numpy(); small[f'{name}_tap{i}_g'] = t[1].numpy()
            else:
                small[f'{name}_tap{i}'] = t.numpy()
    # checksum of the regenerated weights so a drift in the seeded generator is detected
    small['sd_checksum'] = np.array([sum(float(v.double().sum()) for v in sd.values() if v.is_floating_point())])
    np.savez_compressed(os.path.join(HERE, 'small_gen.npz'), **small)

    # ---- isolated FFC units at awkward sizes ------------------------------------------------
    ref_ffc, _ = import_reference()
    units = {}
    g = torch.Generator().manual_seed(3)
    for tag, (b, c, h, w) in dict(e=(2, 6, 8, 12), o=(1, 4, 5, 9), p=(1, 8, 16, 16), q=(1, 4, 10, 7)).items():
        fu = ref_ffc.FourierUnit(c, c).eval()
        with torch.no_grad():
            fu.conv_layer.weight.copy_(torch.randn(fu.conv_layer.weight.shape, generator=g) * 0.3)
            fu.bn.weight.copy_(torch.rand(2 * c, generator=g) + 0.5); fu.bn.bias.copy_(torch.randn(2 * c, generator=g) * 0.2)
            fu.bn.running_mean.copy_(torch.randn(2 * c, generator=g) * 0.1); fu.bn.running_var.copy_(torch.rand(2 * c, generator=g) + 0.5)
            x = torch.randn(b, c, h, w, generator=g)
            y = fu(x)
        for k, v in fu.state_dict().items():
            units[f'fu_{tag}_sd_{k}'] = v.numpy()
        units[f'fu_{tag}_x'] = x.numpy(); units[f'fu_{tag}_y'] = y.numpy()
    # FFC_BN_ACT + FFCResnetBlock with ratio 0.75 on a non-square input
    import torch.nn as nn
    blk = ref_ffc.FFCResnetBlock(16, padding_type='reflect', norm_layer=nn.BatchNorm2d, activation_layer=nn.ReLU,
                                 ratio_gin=0.75, ratio_gout=0.75, enable_lfu=False).eval()
    with torch.no_grad():
        for m in blk.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.weight.copy_(torch.rand(m.weight.shape, generator=g) + 0.5); m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.2)
                m.running_mean.copy_(torch.randn(m.bias.shape, generator=g) * 0.1); m.running_var.copy_(torch.rand(m.bias.shape, generator=g) + 0.5)
        xl, xg = torch.randn(2, 4, 10, 12, generator=g), torch.randn(2, 12, 10, 12, generator=g)
        yl, yg = blk((xl, xg))
        l1, g1 = blk.conv1((xl, xg))
        st = blk.conv1.ffc.convg2g(xg)
    for k, v in blk.state_dict().items():
        units[f'blk_sd_{k}'] = v.numpy()
    units.update(blk_xl=xl.numpy(), blk_xg=xg.numpy(), blk_yl=yl.numpy(), blk_yg=yg.numpy(),
                 blk_c1_l=l1.numpy(), blk_c1_g=g1.numpy(), blk_st=st.numpy())
    np.savez_compressed(os.path.join(HERE, 'ffc_units.npz'), **units)

    # ---- big-lama shape, sampled --------------------------------------------------------------
    cfg = O.BIG_LAMA
    sd = O.make_synthetic_state_dict(cfg, seed=0, calib_hw=64)
    gen = ref_generator(cfg, sd)
    assert sum(p.numel() for p in gen.parameters()) == 50975875
    assert len(gen.state_dict()) == 989
    batch = O.make_synthetic_batch(1, 256, 256, seed=1234)
    x = torch.cat([batch['image'] * (1 - batch['mask']), batch['mask']], 1)
    y, taps = taps_of(gen, x)
    big = dict(y_sample=y[:, :, ::8, ::8].numpy(), y_stat=stat(y),
               sd_checksum=np.array([sum(float(v.double().sum()) for v in sd.values() if v.is_floating_point())]))
    for i, t in taps.items():
        if isinstance(t, tuple):
            if torch.is_tensor(t[1]):
                big[f'tap{i}_g_stat'] = stat(t[1]); big[f'tap{i}_g_sample'] = t[1][:, ::32, ::8, ::8].numpy()
            t = t[0]
        big[f'tap{i}_stat'] = stat(t); big[f'tap{i}_sample'] = t[:, ::16, ::16, ::16].numpy()
    np.savez_compressed(os.path.join(HERE, 'biglama_256.npz'), **big)

    # ---- predict glue: blend + u8 truncation on the small generator ----------------------------
    cfg = O.small_config(ngf=8, n_blocks=2)
    sd = O.make_synthetic_state_dict(cfg, seed=7, calib_hw=32)
    gen = ref_generator(cfg, sd)
    batch = O.make_synthetic_batch(1, 37, 50, seed=5)     # not a multiple of 8 -> symmetric pad
    img, msk = batch['image'][0].numpy(), batch['mask'][0, 0].numpy()
    sys.path.insert(0, REF)
    # reference pad (evaluation/data.py needs cv2 at import; restate the two numpy lines it runs)
    img_p = np.pad(img, ((0, 0), (0, 40 - 37), (0, 56 - 50)), mode='symmetric')
    msk_p = np.pad(msk[None], ((0, 0), (0, 40 - 37), (0, 56 - 50)), mode='symmetric')
    bi, bm = torch.from_numpy(img_p)[None], (torch.from_numpy(msk_p)[None] > 0) * 1
    with torch.no_grad():
        masked = torch.cat([bi * (1 - bm), bm], 1)
        pred = gen(masked)
        inp = bm * pred + (1 - bm) * bi
    cur = inp[0].permute(1, 2, 0).numpy()[:37, :50]
    np.savez_compressed(os.path.join(HERE, 'predict_glue.npz'), image=img, mask=msk, inpainted=cur,
                        u8=np.clip(cur * 255, 0, 255).astype('uint8'))
    print('golden vectors written to', HERE)


if __name__ == '__main__':
    main()
