"""GPU: SSIM kernel (csrc/metrics.hip) against the reference-generated vectors and the oracle."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import eval_oracle as E  # noqa: E402
from oracle import lama_oracle as O  # noqa: E402
from lama_amd import evaluation as EV  # noqa: E402

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(ROOT, 'tests', 'golden', 'ssim.npz'))


@pytest.mark.parametrize('tag', ['a', 'b', 'c'])
def test_ssim_matches_reference_vectors(tag):
    x, y, ws = torch.from_numpy(GOLD[f'{tag}_x']).cuda(), torch.from_numpy(GOLD[f'{tag}_y']).cuda(), int(GOLD[f'{tag}_ws'])
    per = EV.SSIM(window_size=ws, size_average=False)(x, y).cpu().numpy()
    assert np.abs(per - GOLD[f'{tag}_per_image']).max() < 2e-6
    assert abs(float(EV.SSIM(window_size=ws)(x, y)) - float(GOLD[f'{tag}_mean'])) < 2e-6


def test_ssim_configs1_size_and_determinism():
    """8 x 3 x 512 x 512 (the evaluation resolution of BASELINE configs[1]) against the oracle; two runs are bit-identical (fixed-order
    reduction, no atomics)."""
    g = torch.Generator().manual_seed(11)
    x = torch.rand(8, 3, 512, 512, generator=g)
    y = (x + 0.1 * torch.randn(x.shape, generator=g)).clamp(0, 1)
    ref = E.ssim_per_image(y, x, 11)
    m = EV.SSIM(size_average=False)
    a = m(y.cuda(), x.cuda())
    b = m(y.cuda(), x.cuda())
    assert torch.equal(a, b)
    assert float((a.cpu() - ref).abs().max()) < 2e-6
    with pytest.raises(EV.LamaError):
        m(y, x)                                    # CPU tensors: no fallback
