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
