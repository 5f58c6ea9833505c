"""GPU: SSIM evaluator kernel (csrc/metrics.hip) against the reference-generated vectors and the oracle; export (to_jit analogue)."""
import os
import sys

import numpy as np
import pytest
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import eval_oracle as E  # noqa: E402
from oracle import lama_oracle as O  # noqa: E402
from lama_amd import evaluation as EV  # noqa: E402
from lama_amd import export as X  # noqa: E402

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


def test_ssim_score_and_evaluator_on_device():
    from tests.test_eval_oracle import _ToyDataset
    ds = _ToyDataset(n=9, seed=8)
    ev = EV.InpaintingEvaluator(ds, scores={'ssim': EV.SSIMScore()}, bins=4, batch_size=4, device='cuda')
    res = ev.evaluate()
    x = torch.stack([d['image'] for d in ds.items]); y = torch.stack([d['inpainted'] for d in ds.items])
    vals = E.ssim_per_image(y, x, 11).numpy()
    total, per = E.grouped_mean_std(vals, E.area_bins(torch.stack([d['mask'] for d in ds.items]).numpy(), 4))
    assert abs(res[('ssim', 'total')]['mean'] - total['mean']) < 2e-6 and abs(res[('ssim', 'total')]['std'] - total['std']) < 2e-6


def test_export_roundtrip(tmp_path):
    """bin/to_jit.py: checkpoint dir -> one file -> reloaded callable; the printed diff of the reference's self-check is 0 here (same
    kernels, same packed weights), and the reloaded model matches the oracle."""
    cfg = O.small_config(ngf=8, n_blocks=2)
    sd = O.make_synthetic_state_dict(cfg, seed=3, calib_hw=32)
    raw = dict(training_model=dict(kind='default', concat_mask=True), generator=dict(kind='ffc_resnet', **cfg), visualizer=dict(kind='directory'))
    os.makedirs(tmp_path / 'm' / 'models')
    with open(tmp_path / 'm' / 'config.yaml', 'w') as f:
        yaml.safe_dump(raw, f)
    torch.save({'state_dict': {'generator.' + k: v for k, v in sd.items()}}, tmp_path / 'm' / 'models' / 'best.ckpt')
    res = X.export(str(tmp_path / 'm'), str(tmp_path / 'out' / 'small-lama.pt'), size=120)
    assert res['diff'] == 0.0
    w = X.load_exported(str(tmp_path / 'out' / 'small-lama.pt'))
    g = torch.Generator().manual_seed(1)
    img, msk = torch.rand(2, 3, 64, 72, generator=g), (torch.rand(2, 1, 64, 72, generator=g) > 0.7).float()
    out = w(img.cuda(), msk.cuda()).cpu()
    with torch.no_grad():
        pred = O.generator_forward(torch.cat([img * (1 - msk), msk], 1), sd, cfg)
    ref = msk * pred + (1 - msk) * img
    assert float((out - ref).abs().max()) < 2e-4
    assert X.main([]) == 2
