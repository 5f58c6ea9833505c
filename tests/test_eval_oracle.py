"""CPU: the evaluator oracle (oracle/eval_oracle.py) against vectors produced by the reference's own SSIM class
(tests/golden/make_golden_eval.py -> tests/golden/ssim.npz), and the HIP SSIM kernel on the host emulator."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import eval_oracle as E  # noqa: E402
from lama_amd import evaluation as EV  # noqa: E402
from tests.test_kernels_emu import emu_lib  # noqa: E402

GOLD = np.load(os.path.join(ROOT, 'tests', 'golden', 'ssim.npz'))


@pytest.mark.parametrize('tag', ['a', 'b', 'c'])
def test_ssim_oracle_matches_reference_vectors(tag):
    x, y, ws = torch.from_numpy(GOLD[f'{tag}_x']), torch.from_numpy(GOLD[f'{tag}_y']), int(GOLD[f'{tag}_ws'])
    per = E.ssim_per_image(x, y, ws).numpy()
    assert np.abs(per - GOLD[f'{tag}_per_image']).max() < 1e-6
    assert abs(float(per.mean()) - float(GOLD[f'{tag}_mean'])) < 1e-6 or x.shape[0] > 1     # size_average = mean over everything
    if tag == 'c':
        assert abs(float(per[0]) - 1.0) < 1e-6


@pytest.mark.parametrize('tag', ['a', 'b', 'c'])
def test_ssim_hip_kernel_emulated_matches_reference_vectors(tag):
    x, y, ws = torch.from_numpy(GOLD[f'{tag}_x']), torch.from_numpy(GOLD[f'{tag}_y']), int(GOLD[f'{tag}_ws'])
    m = EV.SSIM(window_size=ws, size_average=False)
    m._lib = emu_lib()
    per = m(x, y).numpy()
    assert np.abs(per - GOLD[f'{tag}_per_image']).max() < 2e-6, np.abs(per - GOLD[f'{tag}_per_image']).max()
    m2 = EV.SSIM(window_size=ws, size_average=True)
    m2._lib = emu_lib()
    assert abs(float(m2(x, y)) - float(GOLD[f'{tag}_mean'])) < 2e-6


def test_ssim_hip_kernel_emulated_ragged_and_strided():
    """sizes that are not multiples of the 16 x 32 tile, more tiles than one, a channel-sliced (strided batch) view"""
    lib = emu_lib()
    g = torch.Generator().manual_seed(5)
    big = torch.rand(2, 5, 37, 70, generator=g)
    oth = (big + 0.1 * torch.randn(big.shape, generator=g)).clamp(0, 1)
    from lama_amd import _lib as L
    a, b = big[:, 1:4], oth[:, 1:4]
    out = torch.zeros(2)
    ws = torch.zeros(lib.ssim_workspace_bytes(2, 3, 37, 70) // 4)
    gw = E.gaussian_1d(11).tolist()
    lib.ssim(L.view(big, 1, 3), L.view(oth, 1, 3), 2, gw, out, ws)
    ref = E.ssim_per_image(a.contiguous(), b.contiguous(), 11)
    assert float((out - ref).abs().max()) < 2e-6
    with pytest.raises(L.LamaError):
        lib.ssim(L.view(big, 1, 3), L.view(oth, 1, 3), 2, E.gaussian_1d(10).tolist(), out, ws)      # even window
    with pytest.raises(L.LamaError):
        lib.ssim(L.view(big, 1, 3), L.view(oth, 0, 2), 2, gw, out, ws)                              # channel mismatch
