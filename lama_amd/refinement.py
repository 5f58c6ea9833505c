"""Feature refinement on the HIP path -- mirror of ``saicinpainting/evaluation/refinement.py`` (PR #112; ``refine=True`` of
bin/predict.py:75-79, BASELINE configs[4]).

Same function names, signatures and return values as the reference module.  What differs underneath:

  * no autograd: ``loss.backward()`` (refinement.py:163) is the explicit reverse pass of lama_amd/backward.py (conv / FFT kernels
    with transposed weights + lama_act_bwd / lama_reflect_pad_bwd), ``torch.optim.Adam`` (refinement.py:134,165) is lama_adam_step;
  * kornia's gaussian_blur2d / resize / erosion and the L1 terms are the HIP kernels of lama_amd/csrc/refine.hip
    (lama_gauss5_*, lama_bilinear_*, lama_erode_fwd, lama_threshold_fwd, lama_l1_masked_*); cv2.getStructuringElement is restated;
  * one GPU: the reference splits the resnet blocks over ``gpu_ids`` because a 24 GB card cannot hold the autograd graph at
    1.8 Mpx; with 288 GB of HBM the whole tape (~9 GB at 2048 x 2048) stays on one MI355X, so ``gpu_ids`` / ``devices`` beyond the
    first entry are accepted and ignored ("replicas only").
PyTorch is used for device memory and for slicing / padding copies of the (tiny, once per scale) image and mask tensors only.
"""
from __future__ import annotations

import math
import warnings
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from . import _lib as L
from . import ffc
from ._lib import LamaError, LamaRangeError
from .backward import RearPass


# ----------------------------------------------------------------------------------------------------------------
# helpers on the kernels
# ----------------------------------------------------------------------------------------------------------------

def _lib_of(t: torch.Tensor) -> L.LamaLib:
    lib = getattr(_lib_of, 'override', None)
    if lib is None:
        if not t.is_cuda:
            raise LamaError('lama_amd.refinement runs on an MI355X only (tensor on ' + str(t.device) + ')')
        lib = L.get_lib()
    return lib


def _stream(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream if t.is_cuda else 0


def _c(t: torch.Tensor) -> torch.Tensor:
    return t.contiguous().float()


def _ellipse_kernel(n: int = 15) -> torch.Tensor:
    """cv2.getStructuringElement(cv2.MORPH_ELLIPSE, (n, n)).astype(bool) (refinement.py:128): row i of an n x n ellipse spans
    columns c - dx .. c + dx with dx = round(c * sqrt(1 - dy^2 / r^2)), r = c = n // 2."""
    r = c = n // 2
    k = torch.zeros(n, n)
    for i in range(n):
        dy = i - r
        dx = int(round(c * math.sqrt(max(0.0, (r * r - dy * dy) / float(r * r))))) if r else 0
        k[i, max(c - dx, 0):min(c + dx + 1, n)] = 1.0
    return k


def _pyrdown(im: torch.Tensor, downsize: Optional[tuple] = None) -> torch.Tensor:
    """refinement.py:19-26: gaussian_blur2d(5x5, sigma 1) + bilinear resize to half size."""
    if downsize is None:
        downsize = (im.shape[2] // 2, im.shape[3] // 2)
    assert im.shape[1] == 3, "Expected shape for the input to be (n,3,height,width)"
    lib, im = _lib_of(im), _c(im)
    b = im.shape[0]
    blur = torch.empty_like(im)
    lib.gauss5(L.view(im), L.view(blur), b, _stream(im))
    out = torch.empty(b, 3, int(downsize[0]), int(downsize[1]), device=im.device)
    lib.bilinear(L.view(blur), L.view(out), b, _stream(im))
    return out


def _pyrdown_mask(mask: torch.Tensor, downsize: Optional[tuple] = None, eps: float = 1e-8, blur_mask: bool = True,
                  round_up: bool = True) -> torch.Tensor:
    """refinement.py:28-64."""
    if downsize is None:
        downsize = (mask.shape[2] // 2, mask.shape[3] // 2)
    assert mask.shape[1] == 1, "Expected shape for the input to be (n,1,height,width)"
    lib, mask = _lib_of(mask), _c(mask)
    b, st = mask.shape[0], _stream(mask)
    src = mask
    if blur_mask:
        src = torch.empty_like(mask)
        lib.gauss5(L.view(mask), L.view(src), b, st)
    small = torch.empty(b, 1, int(downsize[0]), int(downsize[1]), device=mask.device)
    lib.bilinear(L.view(src), L.view(small), b, st)
    out = torch.empty_like(small)
    lib.threshold(L.view(small), eps if round_up else 1.0 - eps, L.view(out), b, st)
    return out


def _erode_mask(mask: torch.Tensor, ekernel: Optional[torch.Tensor] = None, eps: float = 1e-8) -> torch.Tensor:
    """refinement.py:66-72."""
    if ekernel is None:
        return mask
    lib, mask = _lib_of(mask), _c(mask)
    b, st = mask.shape[0], _stream(mask)
    er = torch.empty_like(mask)
    lib.erode(L.view(mask), _c(ekernel).to(mask.device), L.view(er), b, 1e4, st)
    out = torch.empty_like(mask)
    lib.threshold(L.view(er), 1.0 - eps, L.view(out), b, st)
    return out


def _masked_l1_terms(pred, target, mask, select_ge: bool) -> Tuple[float, float]:
    """(sum |pred - target|, count) over mask >= 1e-8 (select_ge) or mask < 1e-8."""
    lib = _lib_of(pred)
    acc = torch.zeros(2, dtype=torch.float64, device=pred.device)
    lib.l1_masked(L.view(pred), L.view(target), L.view(mask), 1e-8, select_ge, acc, pred.shape[0], _stream(pred))
    s, n = acc.tolist()
    return s, n


def _l1_loss(pred, pred_downscaled, ref, mask, mask_downscaled, image, on_pred: bool = True) -> float:
    """refinement.py:75-84 (value only; the gradient is lama_l1_masked_bwd inside _infer).  A 1-channel mask stands for
    ``mask.repeat(1,3,1,1)``.  An empty selection contributes 0 here (torch.mean of an empty tensor is NaN in the reference)."""
    s, n = _masked_l1_terms(_c(pred), _c(image), _c(mask), False)
    loss = s / n if n else 0.0
    if on_pred:
        s2, n2 = _masked_l1_terms(_c(pred_downscaled), _c(ref), _c(mask_downscaled), True)
        loss += s2 / n2 if n2 else 0.0
    return loss


# ----------------------------------------------------------------------------------------------------------------
# refinement.py:86-174
# ----------------------------------------------------------------------------------------------------------------

def _state_buffer(z) -> torch.Tensor:
    z1, z2 = z
    buf = ffc._adjacent(z1, z2)
    return (buf if buf is not None else torch.cat([z1, z2], 1)).clone().contiguous()


def _infer(image: torch.Tensor, mask: torch.Tensor, forward_front, forward_rears, ref_lower_res: Optional[torch.Tensor],
           orig_shape: tuple, devices: list, scale_ind: int, n_iters: int = 15, lr: float = 0.002, trace: Optional[dict] = None):
    """refinement.py:86-174.  ``forward_front``: generator.model[0:first_resblock]; ``forward_rears``: a list whose entries are
    RearPass objects (lama_amd/backward.py) -- refine_predict builds ONE covering generator.model[first_resblock:].
    ``trace`` (tests): receives per-iteration losses, the first prediction and the first gradient of z."""
    if len(forward_rears) != 1 or not isinstance(forward_rears[0], RearPass):
        raise LamaError('_infer: forward_rears must be [RearPass] (single device; see the module docstring)')
    rear: RearPass = forward_rears[0]
    lib = _lib_of(image)
    image, mask = _c(image), _c(mask)
    st = _stream(image)
    B, _, H, W = image.shape
    x = torch.empty(B, 4, H, W, device=image.device)
    lib.mask_compose(L.view(image), L.view(mask), L.view(x), B, st)                      # refinement.py:119-120
    with torch.no_grad():
        z = _state_buffer(forward_front(x))                                               # refinement.py:125-126
    ekernel = _ellipse_kernel(15).to(image.device)                                        # refinement.py:128-129
    m_adam, v_adam = torch.zeros_like(z), torch.zeros_like(z)
    oh, ow = int(orig_shape[0]), int(orig_shape[1])
    pred = None
    if ref_lower_res is not None:
        ref = _c(ref_lower_res.to(image.device))
        # the masks do not change during the iterations (refinement.py:150-152 recomputes them every time)
        mask_crop = mask[:, :1, :oh, :ow].contiguous()
        mask_down = _pyrdown_mask(mask_crop, blur_mask=False, round_up=False)
        mask_down = _erode_mask(mask_down, ekernel=ekernel)
        if tuple(mask_down.shape[2:]) != tuple(ref.shape[2:]):
            raise LamaError(f'_infer: reference scale {tuple(ref.shape[2:])} != half of {(oh, ow)}')
        _, n1 = _masked_l1_terms(image, image, mask, False)
        _, n2 = _masked_l1_terms(ref, ref, mask_down, True)
        blur = torch.empty(B, 3, oh, ow, device=image.device)
        pdown = torch.empty_like(ref)
        g_pred = torch.empty(B, 3, H, W, device=image.device)
        g_pdown, g_blur, g_wide = torch.empty_like(ref), torch.empty_like(blur), torch.empty_like(g_pred)
    for idi in range(n_iters):
        pred = rear.forward(z)                                                            # refinement.py:139-146
        if ref_lower_res is None:
            break
        lib.gauss5(L.view(pred), L.view(blur), B, st)                                     # refinement.py:149: _pyrdown(pred[:,:,:oh,:ow])
        lib.bilinear(L.view(blur), L.view(pdown), B, st)
        if trace is not None:
            trace.setdefault('loss', []).append(_l1_loss(pred, pdown, ref, mask, mask_down, image, on_pred=True))
        if idi < n_iters - 1:
            lib.l1_masked_bwd(L.view(pred), L.view(image), L.view(mask), 1e-8, False, (1.0 / n1) if n1 else 0.0, False, L.view(g_pred), B, st)
            lib.l1_masked_bwd(L.view(pdown), L.view(ref), L.view(mask_down), 1e-8, True, (1.0 / n2) if n2 else 0.0, False, L.view(g_pdown), B, st)
            lib.bilinear_bwd(L.view(g_pdown), L.view(g_blur), B, st)
            lib.gauss5_bwd(L.view(g_blur), L.view(g_wide), B, st)
            lib.add(L.view(g_pred), L.view(g_wide), L.view(g_pred), B, st)
            g_z = rear.backward(g_pred)                                                   # refinement.py:163 loss.backward()
            if trace is not None and idi == 0:
                trace['pred0'], trace['g_z'] = pred.clone(), g_z.clone()
            lib.adam_step(z, g_z, m_adam, v_adam, lr, idi + 1, stream=st)                 # refinement.py:165 optimizer.step()
            if trace is not None and trace.get('keep_z'):
                trace.setdefault('z', []).append(z.clone())
                trace.setdefault('g', []).append(g_z.clone())
    inpainted = torch.empty(B, 3, H, W, device=image.device)
    lib.blend(L.view(image), L.view(mask), L.view(pred), L.view(inpainted), B, st)        # refinement.py:171
    return inpainted.detach().cpu()


# ----------------------------------------------------------------------------------------------------------------
# refinement.py:176-226
# ----------------------------------------------------------------------------------------------------------------

def _resize_bilinear(x: torch.Tensor, size: Tuple[int, int]) -> torch.Tensor:
    """kornia.geometry.transform.resize(x, size, interpolation='bilinear', align_corners=False) (refinement.py:200-201)."""
    lib, x = _lib_of(x), _c(x)
    out = torch.empty(x.shape[0], x.shape[1], int(size[0]), int(size[1]), device=x.device)
    lib.bilinear(L.view(x), L.view(out), x.shape[0], _stream(x))
    return out


def _get_image_mask_pyramid(batch: dict, min_side: int, max_scales: int, px_budget: int):
    """refinement.py:176-226."""
    assert batch['image'].shape[0] == 1, "refiner works on only batches of size 1!"
    h, w = batch['unpad_to_size']
    h, w = int(h[0]) if hasattr(h, '__len__') else int(h), int(w[0]) if hasattr(w, '__len__') else int(w)
    image = batch['image'][..., :h, :w]
    mask = batch['mask'][..., :h, :w].float()
    if h * w > px_budget:
        ratio = np.sqrt(px_budget / float(h * w))
        h_orig, w_orig = h, w
        h, w = int(h * ratio), int(w * ratio)
        print(f"Original image too large for refinement! Resizing {(h_orig, w_orig)} to {(h, w)}...")
        image = _resize_bilinear(image, (h, w))
        mask = _resize_bilinear(mask, (h, w))
        # mask[mask > 1e-8] = 1 (refinement.py:202; values below stay as they are -- the scale loop binarises again)
        mask = torch.where(mask > 1e-8, torch.ones_like(mask), mask)
    breadth = min(h, w)
    n_scales = min(1 + int(round(max(0, np.log2(breadth / min_side)))), max_scales)
    ls_images, ls_masks = [_c(image)], [_c(mask)]
    for _ in range(n_scales - 1):
        ls_images.append(_pyrdown(ls_images[-1]))
        ls_masks.append(_pyrdown_mask(ls_masks[-1]))
    return ls_images[::-1], ls_masks[::-1]


def _pad_tensor_to_modulo(img: torch.Tensor, mod: int) -> torch.Tensor:
    """saicinpainting/evaluation/data.py:36-40 (reflect padding at the bottom / right; a copy, no arithmetic)."""
    h, w = img.shape[-2:]
    oh = h if h % mod == 0 else (h // mod + 1) * mod
    ow = w if w % mod == 0 else (w // mod + 1) * mod
    return F.pad(img, pad=(0, ow - w, 0, oh - h), mode='reflect') if (oh != h or ow != w) else img


# ----------------------------------------------------------------------------------------------------------------
# refinement.py:228-314
# ----------------------------------------------------------------------------------------------------------------

def refine_predict(batch: dict, inpainter, gpu_ids: str, modulo: int, n_iters: int, lr: float, min_side: int, max_scales: int,
                   px_budget: int, trace: Optional[list] = None, bwd_precision: Optional[int] = None):
    """refinement.py:228-314: coarse-to-fine refinement of the features after generator.model[:first_resblock].
    Returns the inpainted image [1, 3, H, W] (CPU tensor, like the reference).  ``bwd_precision`` (extension): precision of the
    explicit reverse pass, default the 3-term bf16 split (L.PREC_F32 = exact fp32 MFMA, ~5x slower)."""
    assert not inpainter.training
    assert not getattr(inpainter, 'add_noise_kwargs', None)
    assert inpainter.concat_mask
    ids = [g for g in str(gpu_ids).replace(' ', '').split(',') if g.isdigit()]
    if len(ids) > 1:
        warnings.warn(f'lama_amd.refinement: gpu_ids={gpu_ids!r}: the whole model stays on the first GPU (288 GB of HBM); the others are not used')
    gen = inpainter.generator
    model = gen.model
    first_resblock_ind, found = 0, False
    n_resnet_blocks = 0
    for idl in range(len(model)):                                                         # refinement.py:266-275
        if isinstance(model[idl], ffc.FFCResnetBlock):
            n_resnet_blocks += 1
            found = True
        elif not found:
            first_resblock_ind += 1
    if not n_resnet_blocks:
        raise LamaError('refine_predict: the generator has no resnet blocks')
    device = batch['image'].device
    if not device.type == 'cuda' and getattr(_lib_of, 'override', None) is None:
        raise LamaError('refine_predict: the batch must live on the GPU')
    forward_front = model[0:first_resblock_ind]                                           # refinement.py:281
    ls_images, ls_masks = _get_image_mask_pyramid(batch, min_side, max_scales, px_budget)
    lib = _lib_of(batch['image'])

    def run(rear: RearPass):
        image_inpainted = None
        for ids_, (image, mask) in enumerate(zip(ls_images, ls_masks)):                   # refinement.py:296-312
            orig_shape = image.shape[2:]
            image = _c(_pad_tensor_to_modulo(image, modulo))
            mask = _c(_pad_tensor_to_modulo(mask, modulo))
            mb = torch.empty_like(mask)
            lib.threshold(L.view(mask), 1e-8, L.view(mb), 1, _stream(mask))               # mask[mask >= 1e-8] = 1; mask[mask < 1e-8] = 0
            tr = {} if trace is not None else None
            image_inpainted = _infer(image, mb, forward_front, [rear], image_inpainted, orig_shape, [device], ids_, n_iters, lr, trace=tr)
            image_inpainted = image_inpainted[:, :, :orig_shape[0], :orig_shape[1]]
            if trace is not None:
                tr['out'] = image_inpainted
                trace.append(tr)
        return image_inpainted

    try:
        return run(RearPass(gen, first_resblock_ind) if bwd_precision is None else RearPass(gen, first_resblock_ind, bwd_precision=bwd_precision))
    except LamaRangeError as e:
        if not getattr(gen, 'auto_fallback', True) or gen.precision != L.PREC_F16X3:
            raise
        warnings.warn(f'lama_amd.refinement: {e}; repeating the refinement on the 3-term bf16 split')
        gen.set_precision(L.PREC_BF16X3)
        if trace is not None:
            del trace[:]
        return run(RearPass(gen, first_resblock_ind))
