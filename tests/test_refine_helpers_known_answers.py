"""Known-answer vectors for the four third-party helpers that oracle/refine_oracle.py restates (kornia 0.5.0 / OpenCV are not
installed in this image, so no reference-run fixture can exist).  Every expected value below is a LITERAL worked from the published
source it cites, not computed by the code under test:

  * kornia.filters.gaussian_blur2d(x, (5, 5), (1.0, 1.0))  [kornia/filters/kernels.py ``gaussian``: x = arange(k) - k // 2,
    exp(-x^2 / (2 sigma^2)) / sum; ``get_gaussian_kernel2d`` = outer product; kornia/filters/filter.py ``filter2D``:
    F.pad(mode=border_type='reflect') + grouped F.conv2d, normalized=False]  -- refinement.py:24,54
  * cv2.getStructuringElement(cv2.MORPH_ELLIPSE, (n, n))  [opencv/modules/imgproc/src/morph.dispatch.cpp: r = n / 2, c = n / 2,
    row i: dy = i - r, dx = saturate_cast<int>(c * sqrt((r^2 - dy^2) / r^2)), ones on columns max(c - dx, 0) .. min(c + dx, n - 1)]
    -- refinement.py:131; the 5 x 5 matrix is the one printed in OpenCV's own "Morphological Transformations" tutorial
  * kornia.morphology.erosion(mask, kernel)  [0.5.0: pad with 1.0, one-hot conv per structuring-element position minus (kernel - 1),
    min over positions; >= 0.5.1: 'geodesic' border = pad with max_val = 1e4, unfold, minus the flipped neighbourhood, min]
    -- refinement.py:69.  For masks with values in [0, 1] the two formulations coincide (shown below on random masks); the
    hand-worked cases pin the geometry: a flat structuring element keeps a pixel iff every pixel under the element is 1, and the
    image border does NOT erode.
  * kornia.geometry.transform.resize(x, (h, w), interpolation='bilinear', align_corners=False)  [kornia/geometry/transform/affwarp.py:
    F.interpolate(input, size=size, mode=interpolation, align_corners=align_corners)]  -- refinement.py:209-210; torch's own
    half-pixel rule src = (dst + 0.5) * in / out - 0.5, clamped at 0.

The same vectors are then replayed through the HIP kernels (lama_gauss5_fwd / lama_erode_fwd / lama_bilinear_fwd) on the host SIMT
emulator, so kernel <- oracle <- published algorithm is pinned end to end without the packages.
"""
import numpy as np
import torch
import torch.nn.functional as F

from lama_amd import _lib as L
from lama_amd import refinement as RF
from oracle import refine_oracle as R
from tests.emu import emu_lib

# exp(-x^2 / 2) / sum for x = -2 .. 2 (sum = 2.4837318984903457): float64 literals
G5 = [0.05448868454964294, 0.24420134200323332, 0.4026199468942474, 0.24420134200323332, 0.05448868454964294]

# OpenCV tutorial py_morphological_ops: cv.getStructuringElement(cv.MORPH_ELLIPSE, (5, 5))
ELLIPSE5 = [[0, 0, 1, 0, 0],
            [1, 1, 1, 1, 1],
            [1, 1, 1, 1, 1],
            [1, 1, 1, 1, 1],
            [0, 0, 1, 0, 0]]
# cv.getStructuringElement(cv.MORPH_ELLIPSE, (3, 3)) is the cross (OpenCV documents MORPH_ELLIPSE 3 x 3 == MORPH_CROSS 3 x 3)
ELLIPSE3 = [[0, 1, 0], [1, 1, 1], [0, 1, 0]]
# (15, 15), row by row from the formula: dx = round(sqrt(49 - dy^2)) = 0, 4, 5, 6, 6, 7, 7, 7 for |dy| = 7 .. 0
#   sqrt(13) = 3.61 -> 4, sqrt(24) = 4.90 -> 5, sqrt(33) = 5.74 -> 6, sqrt(40) = 6.32 -> 6, sqrt(45) = 6.71 -> 7, sqrt(48) = 6.93 -> 7
ELLIPSE15 = ['000000010000000',
             '000111111111000',
             '001111111111100',
             '011111111111110',
             '011111111111110',
             '111111111111111',
             '111111111111111',
             '111111111111111',
             '111111111111111',
             '111111111111111',
             '011111111111110',
             '011111111111110',
             '001111111111100',
             '000111111111000',
             '000000010000000']


def _ellipse15():
    return torch.tensor([[float(ch) for ch in row] for row in ELLIPSE15])


def test_gaussian_window_literals():
    g = R.gaussian_kernel1d(5, 1.0)
    assert np.allclose(g.numpy(), G5, atol=1e-7) and abs(float(g.sum()) - 1.0) < 1e-6


def test_gaussian_blur_impulse_and_reflect_border():
    """Interior impulse -> the outer product of the window; an impulse one pixel from the border is counted twice by
    'reflect' (pad = x[1], x[2] mirrored about x[0] WITHOUT repeating the edge), which separates it from replicate / symmetric."""
    x = torch.zeros(1, 1, 9, 9)
    x[0, 0, 4, 4] = 1.0
    y = R.gaussian_blur2d(x)
    assert np.allclose(y[0, 0, 2:7, 2:7].numpy(), np.outer(G5, G5), atol=1e-7) and abs(float(y.sum()) - 1.0) < 1e-6
    r = torch.zeros(1, 1, 5, 8)
    r[0, 0, :, 1] = 1.0                                       # a vertical line at x = 1: 1-D problem along x, constant along y
    y = R.gaussian_blur2d(r)[0, 0, 2]
    #  out[0] = g0 x[2] + g1 x[1] + g2 x[0] + g3 x[1] + g4 x[2] = g1 + g3;  out[1] = g0 x[1] + g1 x[0] + g2 x[1] + ... = g0 + g2
    #  out[2] = g1 x[1] = g1 (x[0] under g0 is 0);  out[3] = g0 x[1] = g0;  out[4] = 0
    exp = [G5[1] + G5[3], G5[0] + G5[2], G5[1], G5[0], 0.0]
    assert np.allclose(y[:5].numpy(), exp, atol=1e-7), y
    lib = emu_lib()
    for src in (x, r):
        out = torch.zeros_like(src)
        lib.gauss5(L.view(src), L.view(out), 1)
        assert torch.allclose(out, R.gaussian_blur2d(src), atol=1e-7)


def test_structuring_elements_literal():
    assert R.ellipse_kernel(5).int().tolist() == ELLIPSE5
    assert R.ellipse_kernel(3).int().tolist() == ELLIPSE3
    assert torch.equal(R.ellipse_kernel(15), _ellipse15())
    assert torch.equal(RF._ellipse_kernel(15).cpu().float(), _ellipse15())      # the product's own restatement (host logic)
    assert RF._ellipse_kernel(5).int().tolist() == ELLIPSE5


def _erosion_kornia_050(x, kernel):
    """kornia 0.5.0 kornia/morphology/basic_operators.py, restated independently of the oracle: pad 1.0, one one-hot filter
    per position of the structuring element (weight 1 where kernel == 1), subtract (kernel - 1), minimum over positions."""
    se_h, se_w = kernel.shape
    se_e = kernel - 1.0
    n = se_h * se_w
    filt = torch.zeros(n, 1, se_h, se_w)
    for i in range(n):
        filt[i, 0, i // se_w, i % se_w] = float(se_e.reshape(-1)[i] >= 0)
    b, c, h, w = x.shape
    out = F.pad(x.reshape(b * c, 1, h, w), [se_w // 2, se_w // 2, se_h // 2, se_h // 2], mode='constant', value=1.0)
    out = F.conv2d(out, filt) - se_e.reshape(1, -1, 1, 1)
    return out.min(dim=1)[0].reshape(b, c, h, w)


def test_erosion_hand_worked():
    lib = emu_lib()
    cross = torch.tensor(ELLIPSE3, dtype=torch.float32)
    m = torch.zeros(1, 1, 9, 9)
    m[0, 0, 2:7, 2:7] = 1.0                                   # 5 x 5 block in the interior
    exp = torch.zeros(9, 9)
    exp[3:6, 3:6] = 1.0                                       # a cross keeps a pixel iff its 4-neighbours are set: the 3 x 3 core
    # ... plus nothing else: the block's edge midpoints have one neighbour outside
    assert torch.equal(R.erosion(m, cross)[0, 0], exp)
    b = torch.zeros(1, 1, 6, 7)
    b[0, 0, 0:4, 0:3] = 1.0                                   # block glued to the top-left corner: the border does not erode it
    expb = torch.zeros(6, 7)
    expb[0:3, 0:2] = 1.0                                      # only the sides facing zeros recede
    assert torch.equal(R.erosion(b, cross)[0, 0], expb)
    for src, e in ((m, exp), (b, expb)):
        out = torch.zeros_like(src)
        lib.erode(L.view(src), cross, L.view(out), 1)
        assert torch.equal(out[0, 0], e)
    # 15 x 15 ellipse: a 21 x 21 block (rows / columns 5 .. 25) shrinks to the centres whose whole ellipse fits.  The ellipse
    # reaches +-7 along its centre row and its centre column (single pixels at dy = +-7), so exactly the 7 x 7 core 12 .. 18 survives
    big = torch.zeros(1, 1, 31, 31)
    big[0, 0, 5:26, 5:26] = 1.0
    core = torch.zeros(31, 31)
    core[12:19, 12:19] = 1.0
    assert torch.equal(R.erosion(big, _ellipse15())[0, 0], core)
    # the empty bounding-box corners of the ellipse: a hole at the block's corner (5, 5) is at (dy, dx) = (-7, -7) of centre
    # (12, 12), outside the element (row dy = -7 is the single pixel dx = 0) -> (12, 12) survives; a hole at (5, 12) is that pixel
    hole_c, hole_t = big.clone(), big.clone()
    hole_c[0, 0, 5, 5] = 0.0
    hole_t[0, 0, 5, 12] = 0.0
    assert torch.equal(R.erosion(hole_c, _ellipse15())[0, 0], core)
    exp_t = core.clone()
    exp_t[12, 12] = 0.0                                       # and only that centre: row dy = -6 spans dx = -4 .. 4 of row r - 6 = 5 -> r = 11, gone anyway
    assert torch.equal(R.erosion(hole_t, _ellipse15())[0, 0], exp_t)
    for src in (big, hole_c, hole_t):
        out = torch.zeros_like(src)
        lib.erode(L.view(src), _ellipse15(), L.view(out), 1)
        assert torch.equal(out, R.erosion(src, _ellipse15()))


def test_erosion_formulations_of_both_kornia_versions_coincide_on_masks():
    g = torch.Generator().manual_seed(31)
    se = _ellipse15()
    for dens in (0.02, 0.2, 0.6):
        m = (torch.rand(2, 1, 40, 52, generator=g) > dens).float()
        m[:, :, 10:30, 12:40] = 1.0
        a, b = R.erosion(m, se), _erosion_kornia_050(m, se)
        assert torch.equal((a >= 1 - 1e-8), (b >= 1 - 1e-8)) and torch.equal(a.clamp(max=1.0), b.clamp(max=1.0))
    gray = torch.rand(1, 1, 24, 24, generator=g)              # gray levels in [0, 1): min over the support, capped by nothing
    assert torch.allclose(R.erosion(gray, se), _erosion_kornia_050(gray, se), atol=0)


def test_bilinear_half_pixel_rule_hand_worked():
    lib = emu_lib()
    ramp = torch.tensor([0.0, 1.0, 2.0, 3.0]).reshape(1, 1, 1, 4).repeat(1, 1, 2, 1)
    down = F.interpolate(ramp, size=(1, 2), mode='bilinear', align_corners=False)
    assert down.flatten().tolist() == [0.5, 2.5]              # src = (dst + 0.5) * 2 - 0.5 = 0.5, 2.5
    two = torch.tensor([0.0, 1.0]).reshape(1, 1, 1, 2)
    up = F.interpolate(two, size=(1, 4), mode='bilinear', align_corners=False)
    assert up.flatten().tolist() == [0.0, 0.25, 0.75, 1.0]    # src = -0.25 (clamped to 0), 0.25, 0.75, 1.25 (clamped to 1)
    three = torch.tensor([0.0, 3.0, 6.0, 9.0, 12.0, 15.0]).reshape(1, 1, 1, 6)
    d3 = F.interpolate(three, size=(1, 4), mode='bilinear', align_corners=False)
    assert np.allclose(d3.flatten().numpy(), [0.75, 5.25, 9.75, 14.25])   # src = 1.5 d + 0.25 -> 0.25, 1.75, 3.25, 4.75 (x 3)
    for src, ref in ((ramp, down), (two, up), (three, d3)):
        out = torch.zeros_like(ref)
        lib.bilinear(L.view(src), L.view(out), 1)
        assert torch.allclose(out, ref, atol=1e-6)
