"""CPU test of lama_amd/refinement.py (mirror of saicinpainting/evaluation/refinement.py) through the host SIMT emulator against
the CPU oracle (oracle/refine_oracle.py: torch autograd + torch.optim.Adam on the generator oracle)."""
import numpy as np
import pytest
import torch

from lama_amd import _lib as L
from lama_amd import ffc as F
from lama_amd import refinement as RF
from lama_amd import trainers
from oracle import lama_oracle as O
from oracle import refine_oracle as R
from tests.emu import emu_lib


@pytest.fixture(scope='module')
def setup():
    cfg = O.small_config(ngf=8, n_blocks=2)
    sd = O.make_synthetic_state_dict(cfg, seed=21, calib_hw=32)
    model = trainers.DefaultInpaintingTrainingModule(dict(generator=dict(kind='ffc_resnet', **cfg)))
    model.load_state_dict({'generator.' + k: v for k, v in sd.items()}, strict=True)
    model.freeze()
    model.generator.set_exec(F._Exec(emu_lib()))
    model.generator.set_precision(L.PREC_F32)
    RF._lib_of.override = emu_lib()
    yield cfg, sd, model
    RF._lib_of.override = None


def test_pyramid_helpers_match_reference_semantics(setup):
    g = torch.Generator().manual_seed(1)
    im = torch.rand(1, 3, 45, 62, generator=g)
    assert torch.allclose(RF._pyrdown(im), R.pyrdown(im), atol=1e-6)
    mask = (torch.rand(1, 1, 45, 62, generator=g) > 0.4).float()
    mask[:, :, 5:35, 10:50] = 1
    for bm, ru in ((True, True), (False, False), (True, False)):
        assert torch.equal(RF._pyrdown_mask(mask, blur_mask=bm, round_up=ru), R.pyrdown_mask(mask, blur_mask=bm, round_up=ru))
    assert torch.equal(RF._ellipse_kernel(15), R.ellipse_kernel(15))
    assert torch.equal(RF._erode_mask(mask, RF._ellipse_kernel(15)), R.erode_mask(mask, R.ellipse_kernel(15)))
    batch = dict(image=torch.rand(1, 3, 72, 96, generator=g), mask=(torch.rand(1, 1, 72, 96, generator=g) > 0.6).float(),
                 unpad_to_size=[torch.tensor([70]), torch.tensor([93])])
    for budget in (100000, 3000):                       # without / with the px_budget resize (refinement.py:195-202)
        ims, mks = RF._get_image_mask_pyramid(batch, min_side=20, max_scales=3, px_budget=budget)
        rims, rmks = R.get_image_mask_pyramid(batch['image'], batch['mask'], (70, 93), 20, 3, budget)
        assert len(ims) == len(rims) == (3 if budget > 3000 else 2)
        for a, b in zip(ims + mks, rims + rmks):
            assert a.shape == b.shape and torch.allclose(a, b, atol=1e-6)


def test_grayscale_mask_gives_the_pyramid_of_the_reference(setup):
    """bin/predict.py:75-81 hands the RAW grayscale mask (PNG levels / 255) to refine_predict -- line 84's binarisation belongs to
    the plain branch.  The pyramid of a gray mask must equal the oracle's on the same gray mask at every scale AFTER the scale loop's
    own threshold (refinement.py:304-305), without and with the px_budget resize."""
    g = torch.Generator().manual_seed(9)
    gray = torch.zeros(1, 1, 72, 96)
    gray[:, :, 12:50, 20:70] = 1.0
    gray[:, :, 10:12, 20:70] = 128 / 255.0                 # anti-aliased rim of a drawn mask
    gray[:, :, 30:40, 70:72] = 3 / 255.0
    batch = dict(image=torch.rand(1, 3, 72, 96, generator=g), mask=gray, unpad_to_size=[torch.tensor([70]), torch.tensor([93])])
    for budget in (100000, 3000):
        ims, mks = RF._get_image_mask_pyramid(batch, min_side=20, max_scales=3, px_budget=budget)
        rims, rmks = R.get_image_mask_pyramid(batch['image'], gray, (70, 93), 20, 3, budget)
        for a, b in zip(mks, rmks):
            assert a.shape == b.shape and torch.allclose(a, b, atol=1e-6)
            assert torch.equal((a >= 1e-8), (b >= 1e-8))
        for a, b in zip(ims, rims):
            assert torch.allclose(a, b, atol=1e-6)


def test_infer_one_scale_matches_autograd_adam(setup):
    """_infer on one scale with a lower-resolution reference: the losses of every iteration, the first gradient and the final
    inpainting against torch autograd + torch.optim.Adam through the oracle (refinement.py:86-174)."""
    cfg, sd, model = setup
    gen = model.generator
    fri = R.first_resblock_index(cfg)
    g = torch.Generator().manual_seed(3)
    H, W, oh, ow = 48, 64, 45, 62
    image = torch.rand(1, 3, H, W, generator=g)
    mask = torch.zeros(1, 1, H, W)
    mask[:, :, 6:40, 8:52] = 1.0                           # large enough to survive the 15 x 15 erosion at half resolution
    ref_low = torch.rand(1, 3, oh // 2, ow // 2, generator=g)
    n_iters, lr = 4, 0.002
    tr_ref, tr = {}, {}
    out_ref = R.infer(image, mask, sd, cfg, ref_low, (oh, ow), n_iters=n_iters, lr=lr, trace=tr_ref)
    from lama_amd.backward import RearPass
    rear = RearPass(gen, fri, bwd_precision=L.PREC_F32)
    out = RF._infer(image, mask, gen.model[0:fri], [rear], ref_low, (oh, ow), ['cpu'], 1, n_iters=n_iters, lr=lr, trace=tr)
    assert len(tr['loss']) == len(tr_ref['loss']) == n_iters
    assert np.allclose(tr['loss'], tr_ref['loss'], rtol=2e-4, atol=1e-6), (tr['loss'], tr_ref['loss'])
    gref = torch.cat([tr_ref['g_z1'], tr_ref['g_z2']], 1)
    assert float((tr['g_z'] - gref).abs().max()) / float(gref.abs().max()) < 1e-3
    assert float((tr['pred0'] - tr_ref['pred0']).abs().max()) < 1e-4
    assert float((out - out_ref).abs().max()) < 2e-3       # Adam normalises the gradient: sign-level differences move z by lr
    # first scale (no reference): plain inference
    out0 = RF._infer(image, mask, gen.model[0:fri], [rear], None, (oh, ow), ['cpu'], 0, n_iters=n_iters, lr=lr)
    ref0 = R.infer(image, mask, sd, cfg, None, (oh, ow), n_iters=n_iters, lr=lr)
    assert float((out0 - ref0).abs().max()) < 1e-4


def test_refine_predict_two_scales(setup):
    cfg, sd, model = setup
    g = torch.Generator().manual_seed(5)
    Hh, Ww = 90, 100
    image = torch.rand(1, 3, 96, 104, generator=g)
    mask = torch.zeros(1, 1, 96, 104)
    mask[:, :, 10:80, 12:90] = 1.0
    batch = dict(image=image, mask=mask, unpad_to_size=[torch.tensor([Hh]), torch.tensor([Ww])])
    trace = []
    out = RF.refine_predict(batch, model, gpu_ids='0,', modulo=8, n_iters=3, lr=0.002, min_side=45, max_scales=2, px_budget=10 ** 6,
                            trace=trace)
    ref = R.refine_predict(image, mask, (Hh, Ww), sd, cfg, modulo=8, n_iters=3, lr=0.002, min_side=45, max_scales=2, px_budget=10 ** 6)
    assert out.shape == ref.shape == (1, 3, Hh, Ww) and len(trace) == 2
    assert float((out - ref).abs().max()) < 5e-3 and float((out - ref).abs().mean()) < 2e-4
