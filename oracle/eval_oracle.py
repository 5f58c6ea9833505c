"""TEST INFRASTRUCTURE ONLY -- CPU oracle of the quality evaluators.

Restates ``saicinpainting/evaluation/losses/ssim.py`` and the grouping arithmetic of ``evaluation/evaluator.py`` /
``losses/base_loss.py`` on torch-CPU / numpy.  Only tests/ may import this module; lama_amd/evaluation.py never does.

Parity pinning: the reference ships no test for these, but ``ssim.py`` imports cleanly in the build container (numpy + torch only), so
``tests/golden/make_golden_eval.py`` runs the reference's own ``SSIM`` class on seeded inputs and commits inputs + outputs as
``tests/golden/ssim.npz``; ``tests/test_eval_oracle.py`` replays them against this restatement.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F


def gaussian_1d(window_size: int, sigma: float = 1.5) -> torch.Tensor:
    """ssim.py:36-40 (float32 values of exp in float64, normalised in float32)."""
    g = torch.tensor([math.exp(-(x - window_size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(window_size)], dtype=torch.float32)
    return g / g.sum()


def ssim_per_image(img1: torch.Tensor, img2: torch.Tensor, window_size: int = 11) -> torch.Tensor:
    """ssim.py:42-71 with size_average=False: [B] means of the SSIM map."""
    c = img1.shape[1]
    g = gaussian_1d(window_size).unsqueeze(1)
    window = g.mm(g.t()).float()[None, None].expand(c, 1, window_size, window_size).contiguous()
    pad = window_size // 2
    conv = lambda x: F.conv2d(x, window, padding=pad, groups=c)          # noqa: E731
    mu1, mu2 = conv(img1), conv(img2)
    mu1_sq, mu2_sq, mu1_mu2 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    sigma1_sq = conv(img1 * img1) - mu1_sq
    sigma2_sq = conv(img2 * img2) - mu2_sq
    sigma12 = conv(img1 * img2) - mu1_mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    ssim_map = ((2 * mu1_mu2 + C1) * (2 * sigma12 + C2)) / ((mu1_sq + mu2_sq + C1) * (sigma1_sq + sigma2_sq + C2))
    return ssim_map.mean(1).mean(1).mean(1)
