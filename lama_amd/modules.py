"""Factory mirroring ``saicinpainting/training/modules/__init__.py:7-19``."""
import logging

from .ffc import FFCResNetGenerator


def make_generator(config, kind, **kwargs):
    logging.info(f'Make generator {kind}')
    if kind == 'ffc_resnet':
        return FFCResNetGenerator(**kwargs)
    if kind in ('pix2pixhd_multidilated', 'pix2pixhd_global'):
        raise NotImplementedError(f'generator kind {kind} is outside the FFC hot path this package accelerates')
    raise ValueError(f'Unknown generator kind {kind}')
