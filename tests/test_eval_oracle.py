"""CPU: the evaluator oracle (oracle/eval_oracle.py) against vectors produced by the reference's own SSIM class
(tests/golden/make_golden_eval.py -> tests/golden/ssim.npz), and the HIP SSIM / evaluator mirror on the host emulator."""
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


class _ToyDataset(torch.utils.data.Dataset):
    def __init__(self, n=7, seed=3):
        g = torch.Generator().manual_seed(seed)
        self.items = []
        for i in range(n):
            img = torch.rand(3, 24, 40, generator=g)
            mask = torch.zeros(1, 24, 40)
            mask[:, : 3 * (i + 1), : 5 * (i + 1)] = 1.0                 # growing hole -> different area bins
            inp = (img + 0.05 * (i + 1) * mask * torch.randn(3, 24, 40, generator=g)).clamp(0, 1)
            self.items.append(dict(image=img, mask=mask, inpainted=inp))

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        return self.items[i]


def test_inpainting_evaluator_mirror_emulated():
    ds = _ToyDataset()
    score = EV.SSIMScore()
    score.score._lib = emu_lib()
    ev = EV.InpaintingEvaluator(ds, scores={'ssim': score}, bins=5, batch_size=3, device='cpu')
    res = ev.evaluate()
    x = torch.stack([d['image'] for d in ds.items]); y = torch.stack([d['inpainted'] for d in ds.items])
    vals = E.ssim_per_image(y, x, 11).numpy()
    groups = E.area_bins(torch.stack([d['mask'] for d in ds.items]).numpy(), 5)
    total, per = E.grouped_mean_std(vals, groups)
    assert abs(res[('ssim', 'total')]['mean'] - total['mean']) < 2e-6 and abs(res[('ssim', 'total')]['std'] - total['std']) < 2e-6
    names = ['0-20%', '20-40%', '40-60%', '60-80%', '80-100%']
    for gidx, st in per.items():
        assert abs(res[('ssim', names[gidx])]['mean'] - st['mean']) < 2e-6
    assert set(k[1] for k in res) == {'total'} | {names[i] for i in per}

    online = EV.make_evaluator(ssim=True, bins=5)
    online.scores['ssim'].score._lib = emu_lib()
    for i in range(0, len(ds), 3):
        items = ds.items[i:i + 3]
        online({k: torch.stack([d[k] for d in items]) for k in ('image', 'mask', 'inpainted')})
    res2 = online.evaluation_end()
    assert abs(res2[('ssim', 'total')]['mean'] - total['mean']) < 2e-6
    assert online.groups == []


def test_fid_lpips_need_the_downloaded_networks():
    from lama_amd._lib import LamaError
    with pytest.raises(LamaError):
        EV.FIDScore()
    with pytest.raises(LamaError):
        EV.LPIPSScore()
    g = torch.Generator().manual_seed(0)
    feat = torch.nn.Linear(12, 6)
    fid = EV.FIDScore(net=lambda b: feat(b.reshape(b.shape[0], -1)))
    a, b = torch.rand(40, 12, generator=g), torch.rand(40, 12, generator=g) + 0.2
    with torch.no_grad():
        fid(a[:20], b[:20]); fid(a[20:], b[20:])
    total, groups = fid.get_value(groups=np.array([0] * 20 + [1] * 19 + [2]))
    with torch.no_grad():
        fa, fb = feat(a).numpy(), feat(b).numpy()
    mu1, mu2, s1, s2 = fa.mean(0), fb.mean(0), np.cov(fa, rowvar=False), np.cov(fb, rowvar=False)
    from scipy import linalg
    ref = ((mu1 - mu2) ** 2).sum() + np.trace(s1) + np.trace(s2) - 2 * np.trace(linalg.sqrtm(s1.dot(s2)).real)
    assert abs(total['mean'] - ref) < 1e-6 * max(1.0, abs(ref)) and np.isnan(groups[2]['mean']) and np.isfinite(groups[0]['mean'])
    assert EV.ssim_fid100_f1({('ssim', 'total'): dict(mean=0.9), ('fid', 'total'): dict(mean=10.0)}) == pytest.approx(2 * 0.9 * 0.9 / (1.8 + 1e-3))


def test_export_roundtrip_emulated(tmp_path):
    """bin/to_jit.py analogue on the host emulator: checkpoint dir -> one file -> reloaded callable, the reference's self-check
    (sum |output - exported output|) is 0, the file holds only tensors and plain containers (torch.load(weights_only=True))."""
    import yaml
    from oracle import lama_oracle as O
    from lama_amd import export as X, ffc as F
    cfg = O.small_config(ngf=8, n_blocks=1)
    sd = O.make_synthetic_state_dict(cfg, seed=3, calib_hw=32)
    os.makedirs(tmp_path / 'm' / 'models')
    with open(tmp_path / 'm' / 'config.yaml', 'w') as f:
        yaml.safe_dump(dict(training_model=dict(kind='default', concat_mask=True), generator=dict(kind='ffc_resnet', **cfg)), f)
    torch.save({'state_dict': {'generator.' + k: v for k, v in sd.items()}}, tmp_path / 'm' / 'models' / 'best.ckpt')
    ex = F._Exec(emu_lib())
    res = X.export(str(tmp_path / 'm'), str(tmp_path / 'o' / 'x.pt'), size=40, executor=ex)
    assert res == dict(diff=0.0, max=0.0)
    w = X.load_exported(str(tmp_path / 'o' / 'x.pt'), executor=ex)
    g = torch.Generator().manual_seed(2)
    img, msk = torch.rand(1, 3, 32, 48, generator=g), (torch.rand(1, 1, 32, 48, generator=g) > 0.6).float()
    out = w(img, msk)
    with torch.no_grad():
        pred = O.generator_forward(torch.cat([img * (1 - msk), msk], 1), sd, cfg)
    assert float((out - (msk * pred + (1 - msk) * img)).abs().max()) < 2e-4
    with pytest.raises(Exception):
        X.load_exported(str(tmp_path / 'm' / 'models' / 'best.ckpt'), executor=ex)      # not an exported file
