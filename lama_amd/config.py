"""Read the training ``config.yaml`` that sits next to a LaMa checkpoint (bin/predict.py:46-48).

The reference parses it with OmegaConf; hydra/omegaconf are not required here: the file is loaded with
``yaml.safe_load`` and the ``${a.b.c}`` interpolations it still contains (bin/train.py:42 saves them
unresolved, e.g. ``ratio_gout: ${generator.resnet_conv_kwargs.ratio_gin}``) are resolved by a small
absolute-path resolver.  ``${env:...}`` entries (never touched when predict_only) are left as-is.
"""
from __future__ import annotations

import re
from typing import Any

import yaml

_INTERP = re.compile(r'^\$\{([A-Za-z0-9_.]+)\}$')


def _lookup(root: dict, path: str):
    cur: Any = root
    for part in path.split('.'):
        cur = cur[part]
    return cur


def resolve(node: Any, root: dict, depth: int = 0) -> Any:
    if depth > 32:
        raise ValueError('interpolation cycle in config')
    if isinstance(node, dict):
        return {k: resolve(v, root, depth) for k, v in node.items()}
    if isinstance(node, list):
        return [resolve(v, root, depth) for v in node]
    if isinstance(node, str):
        m = _INTERP.match(node.strip())
        if m:
            return resolve(_lookup(root, m.group(1)), root, depth + 1)
    return node


def load_train_config(path: str) -> dict:
    with open(path, 'r') as f:
        cfg = yaml.safe_load(f)
    out = dict(cfg)
    for key in ('generator', 'training_model'):
        if key in cfg:
            out[key] = resolve(cfg[key], cfg)
    return out


def generator_kwargs(train_config: dict) -> dict:
    """The kwargs ``make_generator(config, **config.generator)`` receives (trainers/base.py:67)."""
    return dict(train_config['generator'])
