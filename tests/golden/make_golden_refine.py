#!/usr/bin/env python3
"""Golden vectors for BASELINE configs[4]-style refinement on the FULL big-lama generator (18 FFCResnetBlocks).

Runs ``oracle.refine_oracle.refine_predict`` (the torch-CPU autograd + Adam restatement of
``saicinpainting/evaluation/refinement.py:86-174,228-314``) on one seeded 1024 x 1024 image, two scales (512 -> 1024),
``n_iters=15``, ``lr=0.002`` (the reference defaults, ``configs/prediction/default.yaml`` refiner block) and stores

  * (argv[1] = 1024, default, or 2048 = BASELINE configs[4] itself: 3 scales, px_budget 4194304 as bench.py's refine leg)
  * the per-iteration loss curve of every scale (scale 0 has none: ``ref_lower_res is None`` -> one forward),
  * a strided sample (every 8th pixel) + (mean, std, absmax) of the inpainted image after every scale,
  * a checksum of the seeded synthetic state dict (weights are regenerated from the seed at test time).

The generator arithmetic of this oracle is pinned by the reference's own classes (make_golden.py: biglama_256.npz); the four
kornia / OpenCV helpers it restates are pinned by known-answer vectors (tests/test_refine_helpers_known_answers.py).
~10-15 minutes on 8 cores; BUILD CONTAINER or any CPU host (needs no /root/reference).
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

N_ITERS, SEED_SD, SEED_IMG = 15, 0, 77


def make_case(res, seed=SEED_IMG):
    """One smooth-ish random image and a mask of two rectangles + a thin stroke (holes at several scales)."""
    g = torch.Generator().manual_seed(seed)
    low = torch.rand(1, 3, res // 16, res // 16, generator=g)
    image = torch.nn.functional.interpolate(low, size=(res, res), mode='bilinear', align_corners=False)
    image = (image + 0.05 * torch.rand(1, 3, res, res, generator=g)).clamp(0, 1)
    mask = torch.zeros(1, 1, res, res)
    a = res // 16
    mask[:, :, 3 * a:8 * a, 4 * a:11 * a] = 1.0
    mask[:, :, 10 * a:13 * a, 2 * a:6 * a] = 1.0
    mask[:, :, 9 * a:9 * a + a // 4, 7 * a:15 * a] = 1.0
    return image, mask


def stat(t):
    t = t.double()
    return np.array([t.mean().item(), t.std().item(), t.abs().max().item()])


def self_sensitivity(RES, MAX_SCALES, PX_BUDGET, z_noise=2e-6):
    """How far apart are two VALID fp32 evaluations of this very algorithm?  The refinement loss is an L1 (its gradient is sign(pred - image):
    a 1e-5 difference in the prediction flips the sign for ~0.1 % of the pixels) and Adam normalises the gradient (an element whose gradient
    changes sign moves the other way by the full learning rate), so the 15-iteration trajectory is chaotic in the rounding of the forward
    pass.  Measured here by running the ORACLE a second time with the initial features perturbed by a relative 2e-6 (one or two fp32 ulps,
    what a different summation order of the front layers produces) and comparing with the stored golden run: per scale the mean / max
    absolute difference of the inpainted sample and the largest relative difference of the loss curve.  A HIP run cannot be expected to be
    closer to the golden than the oracle is to itself; tests/test_refinement_gpu.py bounds it by a small multiple of these numbers."""
    from oracle import lama_oracle as O
    from oracle import refine_oracle as R
    path = os.path.join(HERE, 'refine_biglama_%d.npz' % RES)
    g = dict(np.load(path))
    cfg = O.BIG_LAMA
    sd = O.make_synthetic_state_dict(cfg, seed=SEED_SD, calib_hw=64)
    image, mask = make_case(RES)
    trace = []
    t0 = time.time()
    R.refine_predict(image, mask, (RES, RES), sd, cfg, modulo=8, n_iters=N_ITERS, lr=0.002, min_side=512, max_scales=MAX_SCALES,
                     px_budget=PX_BUDGET, trace=trace, z_noise=z_noise)
    print('perturbed refine_predict: %.1f s' % (time.time() - t0), flush=True)
    for s, tr in enumerate(trace):
        ref = g[f'out{s}_sample']
        st = tr['out'].shape[-1] // ref.shape[-1]
        d = np.abs(tr['out'][:, :, ::st, ::st].numpy() - ref)
        loss = np.array(tr.get('loss', []), dtype=np.float64)
        rel = float(np.max(np.abs(loss - g[f'loss{s}']) / g[f'loss{s}'])) if len(loss) else 0.0
        g[f'self{s}'] = np.array([d.mean(), d.max(), rel])
        print(f'scale {s}: oracle vs perturbed oracle: out mean-abs {d.mean():.3e} max-abs {d.max():.3e}, loss rel {rel:.3e}', flush=True)
    g['self_z_noise'] = np.array([z_noise])
    np.savez_compressed(path, **g)


def main():
    RES = int(sys.argv[1]) if len(sys.argv) > 1 else 1024      # 1024 -> 2 scales (512, 1024); 2048 -> 3 scales = BASELINE configs[4] as bench.py times it
    MAX_SCALES, PX_BUDGET = (2, 1800000) if RES <= 1024 else (3, 4194304)
    if len(sys.argv) > 2 and sys.argv[2] == 'self':            # second pass: the algorithm's own sensitivity, added to the existing file
        return self_sensitivity(RES, MAX_SCALES, PX_BUDGET)
    from oracle import lama_oracle as O
    from oracle import refine_oracle as R
    cfg = O.BIG_LAMA
    sd = O.make_synthetic_state_dict(cfg, seed=SEED_SD, calib_hw=64)
    image, mask = make_case(RES)
    trace = []
    t0 = time.time()
    out = R.refine_predict(image, mask, (RES, RES), sd, cfg, modulo=8, n_iters=N_ITERS, lr=0.002, min_side=512, max_scales=MAX_SCALES,
                           px_budget=PX_BUDGET, trace=trace)
    print('refine_predict: %.1f s' % (time.time() - t0), flush=True)
    assert len(trace) == MAX_SCALES and out.shape == (1, 3, RES, RES)
    g = dict(res=np.array([RES]), n_iters=np.array([N_ITERS]), seed_sd=np.array([SEED_SD]), seed_img=np.array([SEED_IMG]),
             sd_checksum=np.array([sum(float(v.double().sum()) for v in sd.values() if v.is_floating_point())]))
    for s, tr in enumerate(trace):
        g[f'loss{s}'] = np.array(tr.get('loss', []), dtype=np.float64)
        st = max(8, tr['out'].shape[-1] // 128)              # <= 128 x 128 samples per scale
        g[f'out{s}_sample'] = tr['out'][:, :, ::st, ::st].numpy()
        g[f'out{s}_stride'] = np.array([st])
        g[f'out{s}_stat'] = stat(tr['out'])
        print('scale', s, 'losses', g[f'loss{s}'], flush=True)
    # what refinement changed vs the plain forward at full resolution (so the test can tell "refined" from "not refined")
    with torch.no_grad():
        plain = O.training_module_forward(dict(image=image, mask=mask), {'generator.' + k: v for k, v in sd.items()}, cfg)['inpainted']
    g['sample_stride'] = np.array([max(8, RES // 128)])
    g['plain_sample'] = plain[:, :, ::max(8, RES // 128), ::max(8, RES // 128)].numpy()
    g['refine_minus_plain_meanabs'] = np.array([float((out - plain).abs().mean())])
    np.savez_compressed(os.path.join(HERE, 'refine_biglama_%d.npz' % RES), **g)
    print('written', os.path.join(HERE, 'refine_biglama_%d.npz' % RES), flush=True)


if __name__ == '__main__':
    main()
