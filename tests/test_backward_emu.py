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
