"""TEST INFRASTRUCTURE ONLY -- CPU oracle of the refinement path (BASELINE configs[4]).

Restates ``saicinpainting/evaluation/refinement.py`` (PR #112, "feature refinement") on torch-CPU with autograd, on top
of the generator oracle (oracle/lama_oracle.py).  Only tests/, ``__graft_entry__.smoke()`` and bench.py's cpu_baseline
leg may import this module; the product (lama_amd/refinement.py) never does.

Third-party pieces.  refinement.py calls ``kornia.filters.gaussian_blur2d``, ``kornia.geometry.transform.resize``,
``kornia.morphology.erosion`` and ``cv2.getStructuringElement`` -- neither kornia (reference pins kornia==0.5.0,
requirements.txt) nor opencv is installed in this image, so their published algorithms are restated here.  No reference-run
fixture can be generated offline; the four helpers are instead pinned by **known-answer vectors worked from the cited sources**
(tests/test_refine_helpers_known_answers.py: the 5-tap sigma = 1 window as literals, an impulse / border response of the blur,
OpenCV's documented 5 x 5 ellipse and the 15 x 15 one as a literal matrix, hand-worked erosions incl. the non-eroding border and
the agreement of kornia 0.5.0's conv formulation with the later unfold formulation on masks, hand-worked half-pixel bilinear
cases), and the HIP kernels replay the same vectors.  Everything built on torch (F.interpolate, F.pad, Adam, autograd through
the generator) is the real thing:

  * gaussian_blur2d(x, (5,5), (1,1)): kornia.filters.gaussian -- 1-D window exp(-x^2 / (2 sigma^2)), x = -2..2, normalised to
    sum 1; 2-D kernel = outer product; ``filter2d(border_type='reflect')`` = F.pad(mode='reflect') + depthwise correlation.
  * resize(x, (h,w), interpolation='bilinear', align_corners=False) = F.interpolate(x, size, mode='bilinear',
    align_corners=False) (antialias is off by default).
  * erosion(x, kernel) with the defaults border_type='geodesic', max_val=1e4, origin = kernel centre: the minimum of x over the
    positions where the (flipped) kernel is non-zero, the image being padded with max_val.
  * cv2.getStructuringElement(MORPH_ELLIPSE, (15,15)) (modules/imgproc/src/morph.dispatch.cpp): r = 7, c = 7, row i has ones for
    columns [c - dx, c + dx] with dx = cvRound(c * sqrt((r^2 - dy^2) / r^2)), dy = i - r.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F
from torch import Tensor

from . import lama_oracle as O


# ----------------------------------------------------------------------------------------------
# refinement.py:19-73 helpers
# ----------------------------------------------------------------------------------------------

def gaussian_kernel1d(ksize: int = 5, sigma: float = 1.0) -> Tensor:
    x = torch.arange(ksize, dtype=torch.float32) - ksize // 2
    g = torch.exp(-x * x / (2.0 * sigma * sigma))
    return g / g.sum()


def gaussian_blur2d(x: Tensor, ksize: int = 5, sigma: float = 1.0) -> Tensor:
    """kornia.filters.gaussian_blur2d(x, (k,k), (s,s)) with border_type='reflect'."""
    g = gaussian_kernel1d(ksize, sigma).to(x.dtype)
    k2 = torch.outer(g, g)
    c = x.shape[1]
    p = ksize // 2
    xp = F.pad(x, (p, p, p, p), mode='reflect')
    return F.conv2d(xp, k2[None, None].expand(c, 1, ksize, ksize).contiguous(), groups=c)


def pyrdown(im: Tensor, downsize: Optional[Tuple[int, int]] = None) -> Tensor:
    """refinement.py:19-26."""
    if downsize is None:
        downsize = (im.shape[2] // 2, im.shape[3] // 2)
    assert im.shape[1] == 3
    im = gaussian_blur2d(im)
    return F.interpolate(im, size=downsize, mode='bilinear', align_corners=False)


def pyrdown_mask(mask: Tensor, downsize=None, eps: float = 1e-8, blur_mask: bool = True, round_up: bool = True) -> Tensor:
    """refinement.py:28-64."""
    if downsize is None:
        downsize = (mask.shape[2] // 2, mask.shape[3] // 2)
    assert mask.shape[1] == 1
    if blur_mask:
        mask = gaussian_blur2d(mask)
    mask = F.interpolate(mask, size=downsize, mode='bilinear', align_corners=False)
    if round_up:
        mask = (mask >= eps).to(mask.dtype)
    else:
        mask = (mask >= 1.0 - eps).to(mask.dtype)
    return mask


def ellipse_kernel(n: int = 15) -> Tensor:
    """cv2.getStructuringElement(cv2.MORPH_ELLIPSE, (n, n)).astype(bool) as float (refinement.py:128)."""
    r, c = n // 2, n // 2
    k = torch.zeros(n, n)
    inv_r2 = 1.0 / (r * r) if r else 0.0
    for i in range(n):
        dy = i - r
        if abs(dy) <= r:
            dx = int(round(c * math.sqrt((r * r - dy * dy) * inv_r2)))      # cvRound (ties to even; none occur for n = 15)
            j1, j2 = max(c - dx, 0), min(c + dx + 1, n)
            k[i, j1:j2] = 1.0
    return k


def erosion(x: Tensor, kernel: Tensor, max_val: float = 1e4) -> Tensor:
    """kornia.morphology.erosion defaults: flat structuring element, geodesic border (padded with max_val)."""
    kh, kw = kernel.shape
    py, px = kh // 2, kw // 2
    xp = F.pad(x, (px, px, py, py), mode='constant', value=max_val)
    b, c, h, w = x.shape
    patches = xp.unfold(2, kh, 1).unfold(3, kw, 1)                          # [b, c, h, w, kh, kw]
    neigh = torch.zeros_like(kernel)
    neigh[kernel == 0] = -max_val
    out = patches - neigh.flip((0, 1))[None, None, None, None]
    return out.reshape(b, c, h, w, -1).min(dim=-1)[0]


def erode_mask(mask: Tensor, ekernel: Optional[Tensor] = None, eps: float = 1e-8) -> Tensor:
    """refinement.py:66-72."""
    if ekernel is not None:
        mask = erosion(mask, ekernel)
        mask = (mask >= 1.0 - eps).to(mask.dtype)
    return mask


def l1_loss(pred, pred_downscaled, ref, mask, mask_downscaled, image, on_pred: bool = True) -> Tensor:
    """refinement.py:75-84."""
    loss = torch.mean(torch.abs(pred[mask < 1e-8] - image[mask < 1e-8]))
    if on_pred:
        loss = loss + torch.mean(torch.abs(pred_downscaled[mask_downscaled >= 1e-8] - ref[mask_downscaled >= 1e-8]))
    return loss


def pad_tensor_to_modulo(img: Tensor, mod: int) -> Tensor:
    """saicinpainting/evaluation/data.py:36-40: reflect padding at the bottom / right."""
    h, w = img.shape[-2:]
    oh, ow = O.ceil_modulo(h, mod), O.ceil_modulo(w, mod)
    return F.pad(img, pad=(0, ow - w, 0, oh - h), mode='reflect')


# ----------------------------------------------------------------------------------------------
# refinement.py:86-174  _infer
# ----------------------------------------------------------------------------------------------

def first_resblock_index(cfg: dict) -> int:
    """refinement.py:266-275: index of the first FFCResnetBlock in generator.model."""
    for i, L in enumerate(O.layer_plan(cfg)):
        if L['kind'] == 'resblock':
            return i
    raise ValueError('no resnet block')


def infer(image: Tensor, mask: Tensor, sd: Dict[str, Tensor], cfg: dict, ref_lower_res: Optional[Tensor], orig_shape,
          n_iters: int = 15, lr: float = 0.002, prefix: str = 'model.', trace: Optional[dict] = None, z_noise: float = 0.0) -> Tensor:
    """refinement.py:86-174 on one device.  ``trace`` (optional) receives per-iteration losses and the first-iteration
    gradients of (z1, z2) for the gradient-parity tests.  ``z_noise`` (sensitivity experiments only): relative gaussian perturbation of
    the initial features, e.g. 2e-6 ~ the rounding difference between two valid fp32 evaluations of the front layers."""
    fri = first_resblock_index(cfg)
    masked_image = torch.cat([image * (1 - mask), mask], dim=1)
    mask3 = mask.repeat(1, 3, 1, 1)
    with torch.no_grad():
        z1, z2 = O.run_layers(masked_image, sd, cfg, 0, fri, prefix)
    ekernel = ellipse_kernel(15)
    if z_noise:
        gn = torch.Generator().manual_seed(991)
        z1 = z1 * (1.0 + z_noise * torch.randn(z1.shape, generator=gn))
        z2 = z2 * (1.0 + z_noise * torch.randn(z2.shape, generator=gn))
    z1, z2 = z1.detach().clone().requires_grad_(True), z2.detach().clone().requires_grad_(True)
    opt = torch.optim.Adam([z1, z2], lr=lr)
    pred = None
    for idi in range(n_iters):
        opt.zero_grad()
        pred = O.run_layers((z1, z2), sd, cfg, fri, None, prefix)
        if ref_lower_res is None:
            break
        pred_down = pyrdown(pred[:, :, :orig_shape[0], :orig_shape[1]])
        mask_down = pyrdown_mask(mask3[:, :1, :orig_shape[0], :orig_shape[1]], blur_mask=False, round_up=False)
        mask_down = erode_mask(mask_down, ekernel=ekernel).repeat(1, 3, 1, 1)
        loss = l1_loss(pred, pred_down, ref_lower_res.detach(), mask3, mask_down, image, on_pred=True)
        if trace is not None:
            trace.setdefault('loss', []).append(float(loss))
        if idi < n_iters - 1:
            loss.backward()
            if trace is not None and idi == 0:
                trace['g_z1'], trace['g_z2'] = z1.grad.clone(), z2.grad.clone()
                trace['pred0'] = pred.detach().clone()
            opt.step()
            if trace is not None and trace.get('keep_z'):
                trace.setdefault('z', []).append(torch.cat([z1.detach(), z2.detach()], 1).clone())
                trace.setdefault('g', []).append(torch.cat([z1.grad, z2.grad], 1).clone())
    return (mask3 * pred + (1 - mask3) * image).detach()


# ----------------------------------------------------------------------------------------------
# refinement.py:176-226, 228-314
# ----------------------------------------------------------------------------------------------

def get_image_mask_pyramid(image: Tensor, mask: Tensor, unpad_to_size, min_side: int, max_scales: int, px_budget: int):
    """refinement.py:176-226 (batch of 1)."""
    assert image.shape[0] == 1
    h, w = int(unpad_to_size[0]), int(unpad_to_size[1])
    image, mask = image[..., :h, :w], mask[..., :h, :w]
    if h * w > px_budget:
        ratio = np.sqrt(px_budget / float(h * w))
        h, w = int(h * ratio), int(w * ratio)
        image = F.interpolate(image, size=(h, w), mode='bilinear', align_corners=False)
        mask = F.interpolate(mask, size=(h, w), mode='bilinear', align_corners=False)
        mask = (mask > 1e-8).to(mask.dtype) * 1.0 + (mask <= 1e-8).to(mask.dtype) * mask     # mask[mask>1e-8] = 1
    breadth = min(h, w)
    n_scales = min(1 + int(round(max(0, np.log2(breadth / min_side)))), max_scales)
    ls_images, ls_masks = [image], [mask]
    for _ in range(n_scales - 1):
        ls_images.append(pyrdown(ls_images[-1]))
        ls_masks.append(pyrdown_mask(ls_masks[-1]))
    return ls_images[::-1], ls_masks[::-1]


def refine_predict(image: Tensor, mask: Tensor, unpad_to_size, sd, cfg: dict, modulo: int = 8, n_iters: int = 15, lr: float = 0.002,
                   min_side: int = 512, max_scales: int = 3, px_budget: int = 1800000, prefix: str = 'model.',
                   trace: Optional[list] = None, z_noise: float = 0.0) -> Tensor:
    """refinement.py:228-314 on one device: image [1,3,H,W], mask [1,1,H,W] -> inpainted [1,3,h,w].  ``trace`` (optional list)
    receives one dict per scale: the per-iteration losses of ``infer`` and the scale's inpainted image."""
    ls_images, ls_masks = get_image_mask_pyramid(image, mask, unpad_to_size, min_side, max_scales, px_budget)
    image_inpainted = None
    for img, msk in zip(ls_images, ls_masks):
        orig_shape = img.shape[2:]
        img = pad_tensor_to_modulo(img, modulo)
        msk = pad_tensor_to_modulo(msk, modulo)
        msk = (msk >= 1e-8).to(msk.dtype)
        tr = {} if trace is not None else None
        image_inpainted = infer(img, msk, sd, cfg, image_inpainted, orig_shape, n_iters, lr, prefix, trace=tr, z_noise=z_noise)
        image_inpainted = image_inpainted[:, :, :orig_shape[0], :orig_shape[1]]
        if trace is not None:
            tr['out'] = image_inpainted
            trace.append(tr)
    return image_inpainted
