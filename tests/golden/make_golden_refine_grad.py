#!/usr/bin/env python3
"""Golden vector for the FIRST refinement iteration through the FULL big-lama generator (all 18 FFCResnetBlocks) at 512 x 512:
the prediction ``pred0`` and the gradient of the multi-scale L1 loss with respect to the features (z1, z2) that ``loss.backward()``
leaves in ``z.grad`` (saicinpainting/evaluation/refinement.py:131-165), from ``oracle.refine_oracle`` = torch autograd on the CPU.

Case: make_golden_refine.make_case(512), two scales (256 -> 512, min_side 256), n_iters = 2 (iteration 0 does forward + backward + Adam
step; iteration 1 is the closing forward).  Stored: strided samples of pred0 / the 512-channel gradient (+ full-tensor norms), and the
algorithm's OWN sensitivity -- the same run with the initial features perturbed by a relative 2e-6 (what two valid fp32 evaluations of
the front layers differ by): the loss gradient is sign(pred - image) / N, so pixels whose prediction sits within rounding distance of the
target flip their sign, and the bar of the GPU test (tests/test_refinement_gpu.py::test_first_iteration_gradient_18_blocks) is a small
multiple of that distance.  ~5 minutes on 8 cores; needs no /root/reference (the oracle's control flow is pinned to the reference's own
refinement.py by tests/test_refine_oracle_pin.py)."""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

RES, SEED_SD, SEED_IMG, GS, PS = 512, 0, 77, 4, 8
KW = dict(modulo=8, n_iters=2, lr=0.002, min_side=256, max_scales=2, px_budget=10 ** 8)


def main():
    from make_golden_refine import make_case
    from oracle import lama_oracle as O
    from oracle import refine_oracle as R
    torch.set_num_threads(int(os.environ.get('OMP_NUM_THREADS', '8')))
    cfg = dict(O.BIG_LAMA)
    sd = O.make_synthetic_state_dict(cfg, seed=SEED_SD, calib_hw=64)
    image, mask = make_case(RES, SEED_IMG)
    t0 = time.time()
    tr = []
    R.refine_predict(image, mask, (RES, RES), sd, cfg, trace=tr, **KW)
    g = torch.cat([tr[1]['g_z1'], tr[1]['g_z2']], 1)
    pred0 = tr[1]['pred0']
    print(f'golden run {time.time() - t0:.0f} s; |g| {float(g.norm()):.4e}, losses {tr[1]["loss"]}', flush=True)
    tr2 = []
    R.refine_predict(image, mask, (RES, RES), sd, cfg, trace=tr2, z_noise=2e-6, **KW)
    g2 = torch.cat([tr2[1]['g_z1'], tr2[1]['g_z2']], 1)
    self_g = float((g2 - g).norm() / g.norm())
    self_gs = float((g2 - g)[:, :, ::GS, ::GS].norm() / g[:, :, ::GS, ::GS].norm())
    self_p = float((tr2[1]['pred0'] - pred0).norm() / pred0.norm())
    print(f'self-sensitivity (z perturbed by 2e-6): gradient rel L2 {self_g:.3e} (sample {self_gs:.3e}), pred0 rel L2 {self_p:.3e}', flush=True)
    out = dict(
        g_sample=g[:, :, ::GS, ::GS].numpy(), g_norm=np.array([float(g.norm())]), g_absmax=np.array([float(g.abs().max())]),
        pred0_sample=pred0[:, :, ::PS, ::PS].numpy(), pred0_norm=np.array([float(pred0.norm())]),
        loss=np.asarray(tr[1]['loss'], dtype=np.float64), self_rel=np.array([self_g, self_gs, self_p]),
        strides=np.array([GS, PS]), seed_img=np.array([SEED_IMG]),
        sd_checksum=np.array([sum(float(v.double().sum()) for v in sd.values() if v.is_floating_point())]))
    path = os.path.join(HERE, 'refine_grad_biglama_512.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    main()
