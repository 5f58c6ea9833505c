"""``bin/to_jit.py`` for this engine: freeze a checkpoint directory into ONE self-contained file and check it against the live model.

The reference traces ``JITWrapper(model)`` with ``torch.jit.trace`` (to_jit.py:14-25,55), saves the TorchScript module (:60-61),
reloads it and prints ``(output - jit_output).abs().sum()`` (:63-72).  TorchScript cannot hold HIP launches, and it does not need to:
what ``to_jit`` delivers is "a file that needs neither the training config tree nor Hydra / Lightning to run".  Here that file holds
the resolved generator config, the generator's state dict (reference key names), the precision and the shapes that were captured;
``load_exported`` rebuilds the module from it, packs the weights and replays a captured hipGraph per shape -- the same
``(image, mask) -> inpainted`` callable, and the same self-check.

    python -m lama_amd.export model.path=<dir> [model.checkpoint=best.ckpt] save_path=<file.pt> [precision=f16x3] [size=120]
"""
from __future__ import annotations

import os
from typing import Dict, Optional, Sequence

import torch
import torch.nn as nn

from . import _lib as L
from . import config as lcfg
from . import trainers

FORMAT = 'lama_amd.export.v1'


class JITWrapper(nn.Module):
    """to_jit.py:14-25."""

    def __init__(self, model):
        super().__init__()
        self.model = model

    def forward(self, image, mask):
        return self.model({'image': image, 'mask': mask})['inpainted']


def save_exported(model: trainers.DefaultInpaintingTrainingModule, save_path: str, shapes: Sequence[Sequence[int]] = ()):
    gen = model.generator
    blob = dict(format=FORMAT, abi=L.ABI_VERSION, precision=int(gen.precision),
                config=dict(generator=dict(model.config['generator']),
                            training_model=dict(kind='default', concat_mask=bool(model.concat_mask))),
                state_dict={'generator.' + k: v.detach().cpu() for k, v in gen.state_dict().items()},
                shapes=[[int(v) for v in s] for s in shapes])
    os.makedirs(os.path.dirname(os.path.abspath(save_path)) or '.', exist_ok=True)
    torch.save(blob, save_path)


def load_exported(path: str, device='cuda', executor=None) -> JITWrapper:
    """The exported file -> ``wrapper(image, mask) -> inpainted`` on ``device`` (hipGraph replay per input shape)."""
    blob = torch.load(path, map_location='cpu', weights_only=True)      # tensors + plain containers only: nothing to unpickle
    if not isinstance(blob, dict) or blob.get('format') != FORMAT:
        raise L.LamaError(f'{path}: not a {FORMAT} file')
    model = trainers.make_training_model(blob['config'])
    model.load_state_dict(blob['state_dict'], strict=True)
    model.freeze()
    if executor is not None:                      # host-emulated tests
        model.generator.set_exec(executor)
    else:
        model.to(device)
        model.generator.use_graph = True
    model.generator.set_precision(int(blob['precision']))
    return JITWrapper(model)


def export(model_path: str, save_path: str, checkpoint: str = 'best.ckpt', precision: Optional[int] = None, size: int = 120,
           device='cuda', executor=None, seed: int = 0) -> Dict[str, float]:
    """to_jit.py:28-72; returns {'diff': sum |eager - exported|, 'max': max |...|}."""
    train_config = lcfg.load_train_config(os.path.join(model_path, 'config.yaml'))
    train_config.setdefault('training_model', {})['predict_only'] = True
    model = trainers.load_checkpoint(train_config, os.path.join(model_path, 'models', checkpoint), strict=False, map_location='cpu')
    model.freeze()
    if executor is not None:
        model.generator.set_exec(executor)
        dev = 'cpu'
    else:
        model.to(device)
        dev = device
    if precision is not None:
        model.generator.set_precision(precision)
    wrapper = JITWrapper(model)
    g = torch.Generator().manual_seed(seed)
    image = torch.rand(1, 3, size, size, generator=g).to(dev)         # to_jit.py:46-47
    mask = torch.rand(1, 1, size, size, generator=g).to(dev)
    output = wrapper(image, mask)
    save_exported(model, save_path, shapes=[image.shape])
    print(f'Saving exported model to {save_path}')
    jit_model = load_exported(save_path, device=device, executor=executor)
    print('Checking exported model output...')
    jit_output = jit_model(image, mask)
    diff = float((output - jit_output).abs().sum())
    print(f'diff: {diff}')
    return dict(diff=diff, max=float((output - jit_output).abs().max()))


def main(argv=None) -> int:
    import sys
    args = list(sys.argv[1:] if argv is None else argv)
    kv = dict(a.split('=', 1) for a in args if '=' in a)
    if 'model.path' not in kv or 'save_path' not in kv:
        print(__doc__)
        return 2
    prec = L.PREC_NAMES[kv['precision']] if 'precision' in kv else None
    export(kv['model.path'], kv['save_path'], kv.get('model.checkpoint', 'best.ckpt'), prec, int(kv.get('size', 120)))
    return 0


if __name__ == '__main__':
    raise SystemExit(main())
