"""Model-load / forward path of ``bin/predict.py`` without Lightning.

Mirrors ``saicinpainting/training/trainers/__init__.py:13-30`` (``make_training_model``,
``load_checkpoint``) and the eval branch of ``DefaultInpaintingTrainingModule.forward``
(``trainers/default.py:56-59,67-71,82-86``).  With ``predict_only`` the reference builds nothing but
``self.generator`` (``trainers/base.py:67,73-112``), so that is all this module holds; checkpoint
keys keep their ``generator.`` prefix and everything else in the file (discriminator, evaluator
weights) is ignored exactly as ``strict=False`` does in the reference.
"""
from __future__ import annotations

import logging

import torch
import torch.nn as nn

from . import _lib as L
from .modules import make_generator


class DefaultInpaintingTrainingModule(nn.Module):
    def __init__(self, config: dict, concat_mask=True, predict_only=True, **kwargs):
        super().__init__()
        if not predict_only:
            raise NotImplementedError('lama_amd implements the predict_only (inference) path')
        if kwargs.get('add_noise_kwargs') is not None:
            raise NotImplementedError('add_noise_kwargs is not used by big-lama')
        self.config = config
        self.concat_mask = concat_mask
        gen_cfg = dict(config['generator'])
        self.generator = make_generator(config, **gen_cfg)
        # True (default, the reference's contract): batch['predicted_image'] is a tensor of its own.  False: it is the generator plan's output
        # buffer -- overwritten by the next forward of the same shape -- and the step saves two 25 MB copies at 8 x 512^2 (predict.py and
        # bench.py, which read 'inpainted' only, run this way)
        self.keep_predicted_image = True
        super().train(False)

    def freeze(self):                      # LightningModule.freeze(): eval() + requires_grad_(False)
        for p in self.parameters():
            p.requires_grad_(False)
        return self.eval()

    def train(self, mode: bool = True):
        if mode:
            raise NotImplementedError('inference only')
        return super().train(False)

    def load_state_dict(self, *a, **kw):
        # nn.Module.load_state_dict fills the children through _load_from_state_dict, not through THEIR load_state_dict: the generator's packed
        # weights (BatchNorm folded, MFMA fragment order) must be dropped here, or a second load after a forward would keep computing with the first
        if hasattr(self.generator, '_invalidate'):
            self.generator._invalidate()
        return super().load_state_dict(*a, **kw)

    def on_load_checkpoint(self, state):   # Lightning hook called by load_checkpoint; nothing to restore
        pass

    def forward(self, batch: dict) -> dict:
        img, mask = batch['image'], batch['mask']
        ex = self.generator._exec
        ex.check(img)
        img = img.float().contiguous()
        mask = mask.float().contiguous()          # predict.py:84 makes it int64; img*(1-mask) promotes back
        B, _, H, W = img.shape
        st = ex.stream(img)
        if not self.concat_mask:
            raise NotImplementedError('concat_mask=False is not used by big-lama')
        gen = self.generator
        # the masked image + mask is written straight into the buffer the generator's plan reads (no staging copy inside generator.forward)
        masked = gen.input_buffer((B, 4, H, W), img.device) if hasattr(gen, 'input_buffer') else torch.empty(B, 4, H, W, device=img.device)
        ex.lib.mask_compose(L.view(img), L.view(mask), L.view(masked), B, st)      # default.py:59,67-68
        if hasattr(gen, 'clone_output'):
            keep, gen.clone_output = gen.clone_output, bool(self.keep_predicted_image)
            try:
                pred = gen(masked)                                                # default.py:70
            finally:
                gen.clone_output = keep
        else:
            pred = gen(masked)
        out = torch.empty_like(img)
        ex.lib.blend(L.view(img), L.view(mask), L.view(pred), L.view(out), B, st)  # default.py:71
        batch['predicted_image'] = pred
        batch['inpainted'] = out
        batch['mask_for_losses'] = batch['mask']                                 # default.py:82-84
        return batch


    def forward_u8(self, image_hwc: torch.Tensor, mask: torch.Tensor, sizes, out_u8: torch.Tensor, binarize: bool = True,
                   out_key: str = 'inpainted') -> torch.Tensor:
        """The predict step of bin/predict.py:82-92 on what is on disk (round 6, ABI v110): ``image_hwc`` u8 [B,Hp,Wp,3] and ``mask`` u8 [B,Hp,Wp]
        hold each image's h x w pixels in the top-left corner of its slot, ``sizes`` int32 [B,2] = (h, w) on the device (None: full slots).  The
        first launch does load_image's / 255, pad_img_to_modulo's symmetric padding and predict.py:84's mask > 0 while it composes the
        generator's input (default.py:59,67-68); the last one blends (default.py:71) and writes predict.py:92's clipped u8 HWC image into
        ``out_u8`` [B,Hp,Wp,3] (returned).  Bit-identical to ``forward`` on the fp32 tensors the reference's host code builds + quantize_u8_hwc.
        ``out_key`` = bin/predict.py:86's ``batch[predict_config.out_key]``: 'inpainted' (default.py:71) or 'predicted_image' (default.py:70:
        the generator's output without the blend; the padding columns are cropped by the caller as predict.py:87-90 does)."""
        if out_key not in ('inpainted', 'predicted_image'):
            raise L.LamaError(f"out_key {out_key!r}: the predict_only forward produces 'inpainted' and 'predicted_image' (default.py:70-71)")
        ex = self.generator._exec
        if not ex.injected and not image_hwc.is_cuda:
            raise L.LamaError('lama_amd runs on an MI355X only: input tensor is on ' + str(image_hwc.device) + ' (there is no CPU fallback)')
        if image_hwc.dtype != torch.uint8 or mask.dtype != torch.uint8 or out_u8.dtype != torch.uint8:
            raise L.LamaError('forward_u8: uint8 image / mask / output tensors')
        B, H, W, _ = image_hwc.shape
        if tuple(mask.shape) != (B, H, W) or tuple(out_u8.shape) != (B, H, W, 3) or not (image_hwc.is_contiguous() and mask.is_contiguous() and out_u8.is_contiguous()):
            raise L.LamaError(f'forward_u8: image {tuple(image_hwc.shape)} / mask {tuple(mask.shape)} / out {tuple(out_u8.shape)}: contiguous [B,H,W,3], [B,H,W], [B,H,W,3]')
        if not self.concat_mask:
            raise NotImplementedError('concat_mask=False is not used by big-lama')
        st = ex.stream(image_hwc)
        gen = self.generator
        masked = gen.input_buffer((B, 4, H, W), image_hwc.device) if hasattr(gen, 'input_buffer') else torch.empty(B, 4, H, W, device=image_hwc.device)
        ex.lib.mask_compose_u8(image_hwc, mask, sizes, L.view(masked), B, binarize, st)
        if hasattr(gen, 'clone_output'):
            keep, gen.clone_output = gen.clone_output, False                       # the blend below consumes the plan's own output buffer at once
            try:
                pred = gen(masked)
            finally:
                gen.clone_output = keep
        else:
            pred = gen(masked)
        if out_key == 'predicted_image':
            ex.lib.quantize_u8_hwc(L.view(pred), out_u8, B, H, W, st)              # predict.py:92 on batch['predicted_image']
        else:
            ex.lib.blend_quantize_u8(image_hwc, mask, sizes, L.view(pred), out_u8, B, binarize, st)
        return out_u8


def get_training_model_class(kind):
    if kind == 'default':
        return DefaultInpaintingTrainingModule
    raise ValueError(f'Unknown trainer module {kind}')


def make_training_model(config: dict):
    tm = dict(config.get('training_model', {'kind': 'default'}))
    kind = tm.pop('kind', 'default')
    logging.info(f'Make training model {kind}')
    tm['predict_only'] = True
    allowed = {k: v for k, v in tm.items() if k in ('concat_mask', 'predict_only', 'add_noise_kwargs')}
    return get_training_model_class(kind)(config, **allowed)


def load_checkpoint(train_config: dict, path: str, map_location='cuda', strict=True):
    """trainers/__init__.py:25-30.  ``strict=False`` (as bin/predict.py:58 passes) ignores non-generator keys."""
    model = make_training_model(train_config)
    state = torch.load(path, map_location='cpu', weights_only=False)
    sd = state['state_dict']
    if not strict:
        sd = {k: v for k, v in sd.items() if k.startswith('generator.')}
    res = model.load_state_dict(sd, strict=strict)
    missing = [k for k in res.missing_keys]
    if missing:
        raise RuntimeError(f'checkpoint {path} lacks generator weights: {missing[:5]} ...')
    model.on_load_checkpoint(state)
    if map_location is not None and str(map_location) != 'cpu':
        model.to(map_location)
    return model
