"""CPU test of the explicit reverse pass (lama_amd/backward.py) through the host SIMT emulator: the gradient of a scalar loss w.r.t.
the bottleneck features (z1, z2) of a small FFC generator against torch autograd through the oracle (refinement.py:163)."""
import pytest
import torch

from lama_amd import _lib as L
from lama_amd import ffc as F
from lama_amd.backward import RearPass
from lama_amd.modules import make_generator
from oracle import lama_oracle as O
from oracle import refine_oracle as R
from tests.emu import emu_lib


@pytest.mark.parametrize('hw', [(32, 32), (40, 24)], ids=['pow2_planes', 'generic_dft_planes'])
@pytest.mark.parametrize('bprec', [L.PREC_F32, L.PREC_BF16X3], ids=['bwd_f32', 'bwd_bf16x3'])
def test_rear_gradients_match_autograd(hw, bprec):
    cfg = O.small_config(ngf=8, n_blocks=2)
    sd = O.make_synthetic_state_dict(cfg, seed=11, calib_hw=32)
    gen = make_generator(None, kind='ffc_resnet', **cfg)
    gen.load_state_dict(sd, strict=True)
    gen.set_exec(F._Exec(emu_lib()))
    gen.set_precision(L.PREC_F32)
    fri = R.first_resblock_index(cfg)
    assert fri == 5 and isinstance(gen.model[fri], F.FFCResnetBlock)
    H, W = hw
    batch = O.make_synthetic_batch(1, H, W, seed=4)
    x = torch.cat([batch['image'] * (1 - batch['mask']), batch['mask']], 1)
    with torch.no_grad():
        z1, z2 = O.run_layers(x, sd, cfg, 0, fri)
    z1r, z2r = z1.clone().requires_grad_(True), z2.clone().requires_grad_(True)
    pred_ref = O.run_layers((z1r, z2r), sd, cfg, fri, None)
    gw = torch.randn(pred_ref.shape, generator=torch.Generator().manual_seed(5)) / pred_ref.numel()
    (pred_ref * gw).sum().backward()
    # HIP path: front through the layer sequence, rear through the tape
    zl, zg = gen.model[0:fri](x)
    assert float((zl - z1).abs().max()) < 1e-4 and float((zg - z2).abs().max()) < 1e-4
    rear = RearPass(gen, fri, bwd_precision=bprec)
    z = torch.cat([z1, z2], 1).contiguous()
    pred = rear.forward(z)
    assert float((pred - pred_ref.detach()).abs().max()) < 1e-4
    g = rear.backward(gw.contiguous())
    gref = torch.cat([z1r.grad, z2r.grad], 1)
    scale = float(gref.abs().max())
    tol = 2e-5 if bprec == L.PREC_F32 else 2e-3
    err = float((g - gref).abs().max()) / scale
    assert err < tol, (err, scale)


def test_rear_gradients_with_the_tapes_relu_masks_are_strict():
    """The strict form of the check above (tests/masked_oracle.py): autograd through the oracle with the ReLU masks the HIP forward recorded --
    no mask flip can separate the two gradients, so the bound is rounding only.  (The GPU twin runs all 18 blocks of big-lama this way:
    test_refinement_gpu.py::test_rear_gradients_18_blocks_strict.)"""
    from tests import masked_oracle as MO
    cfg = O.small_config(ngf=8, n_blocks=3)
    sd = O.make_synthetic_state_dict(cfg, seed=13, calib_hw=32)
    gen = make_generator(None, kind='ffc_resnet', **cfg)
    gen.load_state_dict(sd, strict=True)
    gen.set_exec(F._Exec(emu_lib()))
    gen.set_precision(L.PREC_F32)
    fri = R.first_resblock_index(cfg)
    batch = O.make_synthetic_batch(1, 32, 48, seed=9)
    x = torch.cat([batch['image'] * (1 - batch['mask']), batch['mask']], 1)
    with torch.no_grad():
        z1, z2 = O.run_layers(x, sd, cfg, 0, fri)
    rear = RearPass(gen, fri, bwd_precision=L.PREC_F32)
    pred = rear.forward(torch.cat([z1, z2], 1).contiguous())
    gw = torch.randn(pred.shape, generator=torch.Generator().manual_seed(5)) / pred.numel()
    g = rear.backward(gw.contiguous())
    masks = MO.tape_masks(rear)
    assert len(masks) == 3 * 2 * 4 + 3
    pred_ref, gref = MO.rear_gradient(z1, z2, sd, cfg, fri, gw, masks)
    assert float((pred - pred_ref).abs().max()) < 1e-5
    rel = float((g - gref).norm() / gref.norm())
    assert rel < 2e-5 and float((g - gref).abs().max()) / float(gref.abs().max()) < 2e-5, rel
    # ... and the masks matter: with one mask negated the same comparison fails by orders of magnitude
    bad = [m.clone() for m in masks]
    bad[6] = 1.0 - bad[6]
    _, gbad = MO.rear_gradient(z1, z2, sd, cfg, fri, gw, bad)
    assert float((g - gbad).norm() / gref.norm()) > 1e-2
