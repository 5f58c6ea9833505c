"""CPU tests of the refinement kernels (lama_amd/csrc/refine.hip) through the host SIMT emulator against the CPU oracle
(oracle/refine_oracle.py): forward values against the restated kornia / torch ops, adjoints against torch autograd."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from lama_amd import _lib as L
from oracle import refine_oracle as R
from tests.emu import emu_lib


def _g(seed):
    return torch.Generator().manual_seed(seed)


@pytest.mark.parametrize('act', [L.ACT_RELU, L.ACT_SIGMOID, L.ACT_TANH, L.ACT_NONE])
def test_act_bwd_and_add(act):
    lib = emu_lib()
    x = torch.randn(2, 5, 7, 9, generator=_g(1), requires_grad=True)
    y = {L.ACT_RELU: torch.relu, L.ACT_SIGMOID: torch.sigmoid, L.ACT_TANH: torch.tanh, L.ACT_NONE: lambda t: t * 1.0}[act](x)
    g = torch.randn(2, 5, 7, 9, generator=_g(2))
    y.backward(g)
    wide = torch.zeros(2, 8, 7, 9)
    lib.act_bwd(L.view(g), L.view(y.detach().contiguous()), act, L.view(wide, 2, 5), 2)
    assert torch.allclose(wide[:, 2:7], x.grad, atol=1e-6) and float(wide[:, :2].abs().max()) == 0 and float(wide[:, 7:].abs().max()) == 0
    out = torch.zeros(2, 5, 7, 9)
    lib.add(L.view(g), L.view(wide, 2, 5), L.view(out), 2)
    assert torch.equal(out, g + wide[:, 2:7])


@pytest.mark.parametrize('case', [(1, 7, 9), (3, 12, 5), (2, 4, 4), (3, 8, 11)])
def test_reflect_pad_adjoint(case):
    lib = emu_lib()
    pad, H, W = case
    x = torch.randn(2, 3, H, W, generator=_g(3), requires_grad=True)
    xp = F.pad(x, (pad,) * 4, mode='reflect')
    gp = torch.randn(xp.shape, generator=_g(4))
    xp.backward(gp)
    add = torch.randn(2, 3, H, W, generator=_g(5))
    g = torch.zeros(2, 3, H, W)
    lib.reflect_pad_bwd(L.view(gp), None, pad, L.view(g), 2)
    assert torch.allclose(g, x.grad, atol=1e-6)
    lib.reflect_pad_bwd(L.view(gp), L.view(add), pad, L.view(g), 2)
    assert torch.allclose(g, x.grad + add, atol=1e-6)


@pytest.mark.parametrize('case', [(1, 8, 8), (1, 6, 12), (3, 12, 5), (2, 4, 4), (3, 8, 16)])
def test_reflect_pad_adjoint_fused(case):
    """lama_reflect_pad_bwd_fused (v108): fold + the 1x1 path + the identity path (in place) + the activation derivative of the layer upstream,
    against the separate launches it replaces; W % 4 == 0 takes the 16-byte path, the rest the scalar one; views with a channel offset."""
    lib = emu_lib()
    pad, H, W = case
    gp = torch.randn(2, 5, H + 2 * pad, W + 2 * pad, generator=_g(4))
    add1, add2 = torch.randn(2, 5, H, W, generator=_g(5)), torch.randn(2, 7, H, W, generator=_g(6))
    y = torch.randn(2, 7, H, W, generator=_g(7))
    fold = torch.zeros(2, 5, H, W)
    lib.reflect_pad_bwd(L.view(gp), None, pad, L.view(fold), 2)
    for act, d in ((L.ACT_RELU, (y > 0).float()), (L.ACT_TANH, 1 - y * y)):
        s = fold + add1 + add2[:, 2:7]
        g, gm = torch.full((2, 7, H, W), 9.0), torch.full((2, 7, H, W), 9.0)
        lib.reflect_pad_bwd_fused(L.view(gp), L.view(add1), L.view(add2, 2, 5), pad, L.view(y, 2, 5), act, L.view(g, 2, 5), L.view(gm, 2, 5), 2)
        assert torch.equal(g[:, 2:7], s) or torch.allclose(g[:, 2:7], s, atol=1e-6)
        assert torch.allclose(gm[:, 2:7], s * d[:, 2:7], atol=1e-6) and float(g[:, :2].min()) == 9.0 and float(gm[:, :2].min()) == 9.0
    # in place over the identity operand, no mask, one output at a time
    acc = add2.clone()
    lib.reflect_pad_bwd_fused(L.view(gp), None, L.view(acc, 2, 5), pad, None, L.ACT_NONE, L.view(acc, 2, 5), None, 2)
    assert torch.allclose(acc[:, 2:7], fold + add2[:, 2:7], atol=1e-6) and torch.equal(acc[:, :2], add2[:, :2])
    gm = torch.zeros(2, 5, H, W)
    lib.reflect_pad_bwd_fused(L.view(gp), None, None, pad, L.view(y, 0, 5), L.ACT_RELU, None, L.view(gm), 2)
    assert torch.allclose(gm, fold * (y[:, :5] > 0).float(), atol=1e-6)
    with pytest.raises(L.LamaError):
        lib.reflect_pad_bwd_fused(L.view(gp), None, None, pad, None, L.ACT_NONE, None, None, 2)


@pytest.mark.parametrize('case', [(8, 8), (5, 12), (20, 7), (3, 3), (2, 4)])
def test_dgrad_ring_and_fold_from_interior(case):
    """Round 4 (v108): the data gradient of a reflect-padded 3x3 conv without its padded plane -- lama_dgrad_ring_fwd computes the one-pixel
    frame of the zero-padded correlation (exact fp32), lama_reflect_pad_bwd_fused(ring=...) folds it onto the interior (a zero-pad-1 conv).
    Reference: autograd through F.pad(mode='reflect') + conv2d."""
    lib = emu_lib()
    H, W = case
    cin, cout = 64, 128                                      # dgrad: 64 gradient channels in (8 channel groups x 8 channels ahead), 128 data channels out
    wf = torch.randn(cin, cout, 3, 3, generator=_g(8)) * 0.1   # forward conv weight [out = 64, in = 128]
    x = torch.randn(2, cout, H, W, generator=_g(9), requires_grad=True)
    gy = torch.randn(2, cin, H, W, generator=_g(10))
    F.conv2d(F.pad(x, (1,) * 4, mode='reflect'), wf).backward(gy)
    wd = wf.permute(1, 0, 2, 3).flip(2, 3).contiguous()      # w' [128, 64, 3, 3]
    gp = F.conv2d(F.pad(gy, (2,) * 4), wd)                   # the padded plane [2, 128, H + 2, W + 2]
    ring = torch.full((lib.dgrad_ring_bytes(2, cout, H, W) // 4 + 3,), 5.0)
    lib.dgrad_ring(L.view(gy), lib.dgrad_ring_weight(wd), cout, ring, 2)
    r = ring[:-3].view(2, cout, -1)
    assert float(ring[-3:].min()) == 5.0
    assert torch.allclose(r[:, :, :W + 2], gp[:, :, 0], atol=1e-5) and torch.allclose(r[:, :, W + 2:2 * W + 4], gp[:, :, -1], atol=1e-5)
    assert torch.allclose(r[:, :, 2 * W + 4:2 * W + 4 + H], gp[:, :, 1:-1, 0], atol=1e-5)
    assert torch.allclose(r[:, :, 2 * W + 4 + H:], gp[:, :, 1:-1, -1], atol=1e-5)
    interior = gp[:, :, 1:-1, 1:-1].contiguous()
    ident, y = torch.randn(2, cout, H, W, generator=_g(11)), torch.randn(2, cout, H, W, generator=_g(12))
    g, gm = torch.zeros(2, cout, H, W), torch.zeros(2, cout, H, W)
    lib.reflect_pad_bwd_fused(L.view(interior), None, L.view(ident), 1, L.view(y), L.ACT_RELU, L.view(g), L.view(gm), 2, ring=ring)
    assert torch.allclose(g, x.grad + ident, atol=2e-5), float((g - x.grad - ident).abs().max())
    assert torch.allclose(gm, (x.grad + ident) * (y > 0).float(), atol=2e-5)
    with pytest.raises(L.LamaError):
        lib.reflect_pad_bwd_fused(L.view(gp), None, None, 1, None, L.ACT_NONE, L.view(g), None, 2, ring=ring)     # gp must be the interior


@pytest.mark.parametrize('shape', [(16, 24, 16, 24), (16, 24, 13, 21), (9, 8, 5, 6)])
def test_gauss5_crop_fwd_and_adjoint(shape):
    lib = emu_lib()
    XH, XW, H, W = shape
    x = torch.randn(2, 3, XH, XW, generator=_g(6), requires_grad=True)
    ref = R.gaussian_blur2d(x[:, :, :H, :W])
    y = torch.zeros(2, 3, H, W)
    lib.gauss5(L.view(x.detach()), L.view(y), 2)
    assert torch.allclose(y, ref.detach(), atol=1e-6)
    gy = torch.randn(2, 3, H, W, generator=_g(7))
    ref.backward(gy)
    gx = torch.full((2, 3, XH, XW), 9.0)
    lib.gauss5_bwd(L.view(gy), L.view(gx), 2)
    assert torch.allclose(gx, x.grad, atol=1e-6)


@pytest.mark.parametrize('shape', [(16, 24, 8, 12), (13, 21, 6, 10), (9, 7, 4, 3), (10, 12, 17, 20), (8, 8, 8, 8)])
def test_bilinear_fwd_and_adjoint(shape):
    lib = emu_lib()
    H, W, Ho, Wo = shape
    x = torch.randn(2, 3, H, W, generator=_g(8), requires_grad=True)
    ref = F.interpolate(x, size=(Ho, Wo), mode='bilinear', align_corners=False)
    y = torch.zeros(2, 3, Ho, Wo)
    lib.bilinear(L.view(x.detach()), L.view(y), 2)
    assert torch.allclose(y, ref.detach(), atol=1e-6)
    gy = torch.randn(2, 3, Ho, Wo, generator=_g(9))
    ref.backward(gy)
    gx = torch.full((2, 3, H, W), 9.0)
    lib.bilinear_bwd(L.view(gy), L.view(gx), 2)
    assert torch.allclose(gx, x.grad, atol=1e-5)


def test_pyrdown_and_mask_pipeline():
    """refinement.py:19-73: _pyrdown / _pyrdown_mask / _erode_mask restated with the HIP kernels."""
    lib = emu_lib()
    im = torch.rand(1, 3, 21, 30, generator=_g(10))
    blur = torch.zeros_like(im)
    lib.gauss5(L.view(im), L.view(blur), 1)
    down = torch.zeros(1, 3, 10, 15)
    lib.bilinear(L.view(blur), L.view(down), 1)
    assert torch.allclose(down, R.pyrdown(im), atol=1e-6)
    mask = (torch.rand(1, 1, 40, 44, generator=_g(11)) > 0.35).float()
    mask[:, :, 8:30, 10:36] = 1.0
    for blur_mask, round_up in ((True, True), (False, False)):
        src = mask
        if blur_mask:
            src = torch.zeros_like(mask)
            lib.gauss5(L.view(mask), L.view(src), 1)
        small = torch.zeros(1, 1, 20, 22)
        lib.bilinear(L.view(src), L.view(small), 1)
        out = torch.zeros_like(small)
        lib.threshold(L.view(small), 1e-8 if round_up else 1.0 - 1e-8, L.view(out), 1)
        assert torch.equal(out, R.pyrdown_mask(mask, blur_mask=blur_mask, round_up=round_up))
    se = R.ellipse_kernel(15)
    assert se.shape == (15, 15) and float(se.sum()) == 169 and float(se[0].sum()) == 1 and float(se[7].sum()) == 15 and torch.equal(se, se.flip(0, 1))
    er = torch.zeros_like(mask)
    lib.erode(L.view(mask), se, L.view(er), 1)
    assert torch.equal(er, R.erosion(mask, se))
    out = torch.zeros_like(mask)
    lib.threshold(L.view(er), 1.0 - 1e-8, L.view(out), 1)
    assert torch.equal(out, R.erode_mask(mask, se))
    se2 = torch.zeros(3, 5); se2[0, 1] = 1; se2[2, 4] = 1; se2[1, 2] = 1          # asymmetric element: kornia's flip matters
    er2 = torch.zeros_like(mask)
    lib.erode(L.view(mask), se2, L.view(er2), 1)
    assert torch.equal(er2, R.erosion(mask, se2))


def test_masked_l1_and_adam():
    lib = emu_lib()
    g = _g(12)
    pred = torch.rand(1, 3, 9, 14, generator=g, requires_grad=True)
    image = torch.rand(1, 3, 9, 14, generator=g)
    image[0, 1, 2, 3] = float(pred[0, 1, 2, 3])                 # an exact tie: abs'(0) = 0
    mask = (torch.rand(1, 1, 9, 14, generator=g) > 0.5).float()
    m3 = mask.repeat(1, 3, 1, 1)
    loss = torch.mean(torch.abs(pred[m3 < 1e-8] - image[m3 < 1e-8]))
    loss.backward()
    acc = torch.zeros(2, dtype=torch.float64)
    lib.l1_masked(L.view(pred.detach()), L.view(image), L.view(mask), 1e-8, False, acc, 1)
    assert int(acc[1]) == int((m3 < 1e-8).sum()) and abs(float(acc[0] / acc[1]) - float(loss)) < 1e-6
    gbuf = torch.full((1, 3, 9, 14), 5.0)
    lib.l1_masked_bwd(L.view(pred.detach()), L.view(image), L.view(mask), 1e-8, False, float(1.0 / acc[1]), False, L.view(gbuf), 1)
    assert torch.allclose(gbuf, pred.grad, atol=1e-7)
    lib.l1_masked_bwd(L.view(pred.detach()), L.view(image), L.view(m3), 1e-8, True, 0.25, True, L.view(gbuf), 1)   # accumulate, 3-channel mask, >=
    assert torch.allclose(gbuf, pred.grad + 0.25 * torch.sign(pred.detach() - image) * (m3 >= 1e-8), atol=1e-7)
    # Adam against torch.optim.Adam, three steps
    p0 = torch.randn(1000, generator=g)
    pt = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([pt], lr=0.002)
    ph, m, v = p0.clone(), torch.zeros(1000), torch.zeros(1000)
    for step in range(1, 4):
        gr = torch.randn(1000, generator=g) * (10.0 ** (step - 2))
        opt.zero_grad(); pt.grad = gr.clone(); opt.step()
        lib.adam_step(ph, gr, m, v, 0.002, step)
        assert torch.allclose(ph, pt.detach(), atol=1e-6, rtol=1e-5), step
