"""``dataset.scale_factor`` (evaluation/data.py:42-55,74-77: cv2.resize with INTER_AREA on the image, INTER_NEAREST on the mask).

cv2 is not in this image, so lama_amd.predict.scale_image is a restatement of OpenCV's resize.cpp and its parity with cv2 itself is UNPINNED
(no vectors could be generated).  What is checked here: known answers that follow from the definition of the two interpolations (box means,
exact area integrals in float64, nearest taps, dsize rounding), and that the predict loop feeds the rescaled fp32 tensors through the step
(host SIMT emulator) with the on-disk contract of bin/predict.py -- results at the rescaled size, within one u8 level of the oracle's predict
loop on the same rescaled inputs."""
import math
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from lama_amd import predict as P  # noqa: E402


def _area_integral(a, factor):
    """INTER_AREA by its definition, float64: destination pixel d covers [d * s, (d + 1) * s) of the source axis (s = 1 / factor), cut at the
    border, every source pixel weighted by its overlap / covered length -- separable."""
    def weights(ssize, dsize, s):
        w = np.zeros((dsize, ssize))
        for d in range(dsize):
            lo, hi = d * s, min((d + 1) * s, ssize)
            for x in range(int(math.floor(lo)), int(math.ceil(hi))):
                ov = min(hi, x + 1) - max(lo, x)
                if ov > 0:
                    w[d, x] = ov
            w[d] /= w[d].sum()
        return w
    H, W = a.shape[-2:]
    dh, dw = P.scaled_size(H, W, factor)
    wy, wx = weights(H, dh, 1.0 / factor), weights(W, dw, 1.0 / factor)
    return np.einsum('yh,...hw,xw->...yx', wy, a.astype(np.float64), wx)


def test_scaled_size_rounds_half_to_even_like_saturate_cast():
    assert P.scaled_size(5, 7, 0.5) == (2, 4)          # 2.5 -> 2, 3.5 -> 4
    assert P.scaled_size(100, 136, 0.37) == (37, 50)
    assert P.scaled_size(9, 11, 1.5) == (14, 16)       # 13.5 -> 14, 16.5 -> 16


def test_integer_shrink_is_the_box_mean():
    b = np.arange(48, dtype=np.float32).reshape(3, 4, 4) / 7
    got = P.scale_image(b, 0.5)
    want = b.reshape(3, 2, 2, 2, 2).transpose(0, 1, 3, 2, 4).reshape(3, 2, 2, 4)
    want = (((want[..., 0] + want[..., 1]) + want[..., 2]) + want[..., 3]) * np.float32(0.25)   # row-major sum of the box, * 1 / area
    assert got.dtype == np.float32 and np.array_equal(got, want)
    c = np.arange(81, dtype=np.float32).reshape(1, 9, 9)
    assert np.allclose(P.scale_image(c, 1 / 3.0)[0], c[0].reshape(3, 3, 3, 3).mean((1, 3)), rtol=1e-6)
    # odd sizes: dsize = round(size / 2) may take one more pixel than there are whole boxes; it averages what is inside the image
    d = np.arange(35, dtype=np.float32).reshape(1, 5, 7)
    got = P.scale_image(d, 0.5)
    assert got.shape == (1, 2, 4)
    assert got[0, 0, 3] == np.float32((d[0, 0, 6] + d[0, 1, 6]) / 2)


@pytest.mark.parametrize('factor', [0.37, 0.6, 0.75, 0.9])
def test_fractional_shrink_is_the_area_integral(factor):
    rng = np.random.default_rng(1)
    a = rng.random((3, 23, 31), dtype=np.float32)
    got = P.scale_image(a, factor)
    want = _area_integral(a, factor)
    assert got.shape == want.shape and got.dtype == np.float32
    assert np.abs(got - want).max() < 2e-6


def test_constant_image_and_identity():
    c = np.full((3, 9, 11), 0.25, np.float32)
    for f in (0.3, 0.5, 0.77, 1.3, 2.0):
        assert np.abs(P.scale_image(c, f) - 0.25).max() < 1e-7
    rng = np.random.default_rng(2)
    a = rng.random((3, 6, 5), dtype=np.float32)
    assert np.array_equal(P.scale_image(a, 1.0), a)


def test_enlarging_with_inter_area_is_linear_with_area_coefficients():
    a = np.array([[[0.0, 1.0, 2.0, 3.0]]], np.float32)                    # one row: the x coefficients alone
    got = P.scale_image(a, 2.0)[0, 0]
    # sx = floor(dx / 2); fx = (dx + 1) - (sx + 1) * 2 -> 0 for even dx (<= 0), for odd dx: 0 -> frac(0) = 0: an integer factor repeats pixels
    assert np.array_equal(got, np.repeat(a[0, 0], 2))
    got = P.scale_image(a, 1.5)[0, 0]                                     # dsize 6, scale 2/3: sx = 0 0 1 2 2 3, fx = 0 .5 0 0 .5 0
    assert np.allclose(got, [0.0, 0.5, 1.0, 2.0, 2.5, 3.0], atol=1e-6)


def test_nearest_taps_floor_of_the_source_coordinate():
    m = np.arange(30, dtype=np.float32).reshape(1, 5, 6)
    got = P.scale_image(m, 0.5, interpolation='nearest')
    assert np.array_equal(got[0], m[0][[0, 2]][:, [0, 2, 4]])
    got = P.scale_image(m, 1.7, interpolation='nearest')
    ys = np.minimum(np.floor(np.arange(8) / 1.7).astype(int), 4)          # dsize = round(5 * 1.7) = 8 (8.5 -> 8), round(6 * 1.7) = 10
    xs = np.minimum(np.floor(np.arange(10) / 1.7).astype(int), 5)
    assert got.shape == (1, 8, 10) and np.array_equal(got[0], m[0][ys][:, xs])
    hole = np.zeros((1, 8, 8), np.float32)
    hole[0, 2:6, 2:6] = 1.0
    assert set(np.unique(P.scale_image(hole, 0.6, interpolation='nearest'))) <= {0.0, 1.0}   # a mask stays binary


def test_option_parsing():
    cfg = P.parse_overrides(['model.path=m', 'indir=i', 'outdir=o', 'dataset.scale_factor=0.5'])
    assert cfg['dataset.scale_factor'] == 0.5
    assert P.parse_overrides(['model.path=m', 'indir=i', 'outdir=o']).get('dataset.scale_factor') is None
    with pytest.raises(SystemExit):
        P.parse_overrides(['model.path=m', 'indir=i', 'outdir=o', 'dataset.scale_factor=0'])
    with pytest.raises(SystemExit):
        P.parse_overrides(['model.path=m', 'indir=i', 'outdir=o', 'dataset.scale_factor=half'])


def test_load_item_rescales_before_padding(tmp_path):
    from PIL import Image
    rng = np.random.RandomState(0)
    img = rng.randint(0, 256, (37, 50, 3)).astype('uint8')
    mask = np.zeros((37, 50), 'uint8')
    mask[9:20, 16:33] = 255
    Image.fromarray(img).save(tmp_path / 'a.png')
    Image.fromarray(mask).save(tmp_path / 'a_mask.png')
    image, m, hw = P.load_item(str(tmp_path / 'a_mask.png'), str(tmp_path / 'a.png'), 8, 0.6)
    assert hw == (22, 30) and image.shape == (3, 24, 32) and m.shape == (1, 24, 32)      # unpad_to_size = the RESCALED size (data.py:79)
    want = P.scale_image(img.transpose(2, 0, 1).astype('float32') / 255, 0.6)
    assert np.array_equal(image[:, :22, :30], want)
    assert np.array_equal(image[:, 22:, :30], want[:, :-3:-1])                           # symmetric padding of the rescaled image
    assert set(np.unique(m)) <= {0.0, 1.0}


def test_predict_loop_with_scale_factor_on_the_emulator(tmp_path):
    """The whole loop: PNG header sizes -> rescaled bucket shapes, host rescale + padding on the pool, HostFedStep in its fp32 form, results
    cropped to the rescaled size.  Expected: the oracle's predict loop on the same rescaled image / mask."""
    from PIL import Image
    from oracle import lama_oracle as O
    from tests.test_dist_gloo import _build_model, _make_dataset
    indir, _ = _make_dataset(str(tmp_path))
    out = str(tmp_path / 'out')
    model, sd, cfg = _build_model()
    items = P.list_dataset(indir, '.png')
    factor = 0.8
    assert P.predict(model, items, indir, out, pad_mod=8, batch_size=2, device='cpu', io_threads=2, scale_factor=factor) == len(items)
    for mask_path, img_path in items:
        rel = os.path.splitext(mask_path[len(indir):])[0] + '.png'
        got = np.array(Image.open(os.path.join(out, rel)))
        image = P.scale_image(O.load_image(img_path, 'RGB'), factor)
        mask = P.scale_image(O.load_image(mask_path, 'L')[None], factor, interpolation='nearest')[0]
        _, u8 = O.predict_one(image, mask, sd, cfg)
        assert got.shape == u8.shape == (image.shape[1], image.shape[2], 3), rel
        assert np.abs(got.astype(int) - u8.astype(int)).max() <= 1, rel
        assert (got != u8).mean() < 0.02
