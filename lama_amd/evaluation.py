"""Quality evaluators: the reference's ``saicinpainting/evaluation`` surface for scoring inpainting results.

Mirrors, name by name:

  * ``losses/ssim.py``       ``SSIM``                    -> one HIP launch pair (``lama_ssim_fwd``, csrc/metrics.hip)
  * ``losses/base_loss.py``  ``get_groupings``, ``EvaluatorScore``, ``PairwiseScore``, ``SSIMScore``, ``FIDScore`` (Frechet distance on
                             activations the CALLER's feature extractor produces), ``LPIPSScore`` (same: a caller-supplied distance)
  * ``evaluator.py``         ``InpaintingEvaluator``, ``InpaintingEvaluatorOnline``, ``ssim_fid100_f1``, ``lpips_fid100_f1``
  * ``__init__.py``          ``make_evaluator``
  * ``data.py``              ``PrecomputedInpaintingResultsDataset`` (for ``bin/evaluate_predicts.py`` -> ``python -m lama_amd.evaluation``)

What is NOT here and why: LPIPS needs the pretrained VGG-16 + linear heads (``models/lpips_models/vgg.pth``), FID the pretrained
Inception-v3 (``pt_inception-2015-12-05``), the segmentation-aware scores the ADE20k segmentation network -- all downloaded weights
that do not exist offline, so there is nothing to be in parity WITH.  ``LPIPSScore`` / ``FIDScore`` therefore take the network as a
callable (any ``nn.Module``) and implement everything around it (accumulation, grouping, the Frechet distance).
"""
from __future__ import annotations

import math
import os
from typing import Callable, Dict, Optional

import numpy as np
import torch
import torch.nn as nn

from . import _lib as L
from ._lib import LamaError


# ----------------------------------------------------------------------------------------------------
# losses/ssim.py
# ----------------------------------------------------------------------------------------------------

class SSIM(nn.Module):
    """ssim.py:7-74.  ``forward(img1, img2)`` on device tensors [B, C, H, W] fp32: the mean SSIM (``size_average``) or one value per
    image; the five depthwise window convolutions and the map never reach HBM (csrc/metrics.hip)."""

    def __init__(self, window_size: int = 11, size_average: bool = True):
        super().__init__()
        self.window_size = window_size
        self.size_average = size_average
        self.channel = 1
        self.register_buffer('window', self._create_window(window_size, self.channel))
        self._lib: Optional[L.LamaLib] = None

    def _gaussian(self, window_size: int, sigma: float) -> torch.Tensor:                 # ssim.py:36-40
        gauss = torch.Tensor([np.exp(-(x - (window_size // 2)) ** 2 / float(2 * sigma ** 2)) for x in range(window_size)])
        return gauss / gauss.sum()

    def _create_window(self, window_size: int, channel: int) -> torch.Tensor:            # ssim.py:42-45 (kept for state-dict shape)
        w1 = self._gaussian(window_size, 1.5).unsqueeze(1)
        w2 = w1.mm(w1.t()).float().unsqueeze(0).unsqueeze(0)
        return w2.expand(channel, 1, window_size, window_size).contiguous()

    def forward(self, img1: torch.Tensor, img2: torch.Tensor) -> torch.Tensor:
        assert len(img1.shape) == 4
        if img1.shape != img2.shape:
            raise LamaError(f'SSIM: shapes differ {tuple(img1.shape)} vs {tuple(img2.shape)}')
        lib = self._lib or L.get_lib()
        if not lib.host_emulated and not img1.is_cuda:
            raise LamaError('SSIM runs on the GPU: move the images to the device (there is no CPU path)')
        a, b = img1.contiguous().float(), img2.contiguous().float()
        B, Cn, H, W = a.shape
        out = torch.empty(B, dtype=torch.float32, device=a.device)
        ws = torch.empty(max(1, lib.ssim_workspace_bytes(B, Cn, H, W) // 4), dtype=torch.float32, device=a.device)
        g = self._gaussian(self.window_size, 1.5).tolist()
        lib.ssim(L.view(a), L.view(b), B, g, out, ws, stream=L.LamaLib.stream_of(a))
        return out.mean() if self.size_average else out

    def _load_from_state_dict(self, *args, **kwargs):                                     # ssim.py:73-74: the window is never loaded
        return


# ----------------------------------------------------------------------------------------------------
# losses/base_loss.py
# ----------------------------------------------------------------------------------------------------

def get_groupings(groups) -> Dict[int, np.ndarray]:
    """base_loss.py:21-37: {group id: indices of its elements}."""
    label_groups, count_groups = np.unique(groups, return_counts=True)
    indices = np.argsort(groups, kind='stable')
    grouping, cur = {}, 0
    for label, count in zip(label_groups, count_groups):
        grouping[label] = indices[cur:cur + count]
        cur += count
    return grouping


class EvaluatorScore(nn.Module):
    """base_loss.py:40-51."""

    def forward(self, pred_batch, target_batch, mask):
        raise NotImplementedError

    def get_value(self, groups=None, states=None):
        raise NotImplementedError

    def reset(self):
        raise NotImplementedError


class PairwiseScore(EvaluatorScore):
    """base_loss.py:54-89: one value per sample; mean / std overall and per group."""

    def __init__(self):
        super().__init__()
        self.individual_values = None

    def get_value(self, groups=None, states=None):
        individual_values = torch.cat(states, dim=-1).reshape(-1).cpu().numpy() if states is not None else self.individual_values
        total_results = {'mean': individual_values.mean(), 'std': individual_values.std()}
        if groups is None:
            return total_results, None
        group_results = {}
        for label, index in get_groupings(groups).items():
            scores = individual_values[index]
            group_results[label] = {'mean': scores.mean(), 'std': scores.std()}
        return total_results, group_results

    def reset(self):
        self.individual_values = []


class SSIMScore(PairwiseScore):
    """base_loss.py:92-103."""

    def __init__(self, window_size: int = 11):
        super().__init__()
        self.score = SSIM(window_size=window_size, size_average=False).eval()
        self.reset()

    def forward(self, pred_batch, target_batch, mask=None):
        batch_values = self.score(pred_batch, target_batch)
        self.individual_values = np.hstack([self.individual_values, batch_values.detach().cpu().numpy()])
        return batch_values


class LPIPSScore(PairwiseScore):
    """base_loss.py:106-118 around a caller-supplied perceptual distance ``net(pred, target) -> [B]`` (the reference's PerceptualLoss
    needs the downloaded VGG weights)."""

    def __init__(self, net: Optional[Callable] = None, **_):
        super().__init__()
        if net is None:
            raise LamaError('LPIPSScore needs the pretrained perceptual network (models/lpips_models/vgg.pth in the reference): pass it '
                            'as net=callable(pred, target) -> per-sample distances')
        self.score = net
        self.reset()

    def forward(self, pred_batch, target_batch, mask=None):
        batch_values = self.score(pred_batch, target_batch).flatten()
        self.individual_values = np.hstack([self.individual_values, batch_values.detach().cpu().numpy()])
        return batch_values


def fid_calculate_activation_statistics(act: np.ndarray):
    """base_loss.py:121-124."""
    return np.mean(act, axis=0), np.cov(act, rowvar=False)


def calculate_frechet_distance(activations_pred: np.ndarray, activations_target: np.ndarray, eps: float = 1e-6) -> float:
    """base_loss.py:127-153: |mu1 - mu2|^2 + Tr(S1 + S2 - 2 sqrt(S1 S2))."""
    from scipy import linalg
    mu1, sigma1 = fid_calculate_activation_statistics(activations_pred)
    mu2, sigma2 = fid_calculate_activation_statistics(activations_target)
    diff = mu1 - mu2
    covmean, _ = linalg.sqrtm(sigma1.dot(sigma2), disp=False)
    if not np.isfinite(covmean).all():
        offset = np.eye(sigma1.shape[0]) * eps
        covmean = linalg.sqrtm((sigma1 + offset).dot(sigma2 + offset))
    if np.iscomplexobj(covmean):
        if not np.allclose(np.diagonal(covmean).imag, 0, atol=1e-2):
            raise ValueError('Imaginary component {}'.format(np.max(np.abs(covmean.imag))))
        covmean = covmean.real
    return float(diff.dot(diff) + np.trace(sigma1) + np.trace(sigma2) - 2 * np.trace(covmean))


class FIDScore(EvaluatorScore):
    """base_loss.py:156-218 around a caller-supplied feature extractor ``net(batch) -> [B, dims]`` (the reference's InceptionV3 needs
    the downloaded weights)."""

    def __init__(self, net: Optional[Callable] = None, eps: float = 1e-6, **_):
        super().__init__()
        if net is None:
            raise LamaError('FIDScore needs the pretrained Inception-v3 feature extractor: pass it as net=callable(batch) -> [B, dims]')
        self.model = net
        self.eps = eps
        self.reset()

    def forward(self, pred_batch, target_batch, mask=None):
        ap, at = self._get_activations(pred_batch), self._get_activations(target_batch)
        self.activations_pred.append(ap.detach().cpu())
        self.activations_target.append(at.detach().cpu())
        return ap, at

    def get_value(self, groups=None, states=None):
        ap, at = zip(*states) if states is not None else (self.activations_pred, self.activations_target)
        ap, at = torch.cat(ap).cpu().numpy(), torch.cat(at).cpu().numpy()
        total_results = dict(mean=calculate_frechet_distance(ap, at, eps=self.eps))
        if groups is None:
            group_results = None
        else:
            group_results = {}
            for label, index in get_groupings(groups).items():
                if len(index) > 1:
                    group_results[label] = dict(mean=calculate_frechet_distance(ap[index], at[index], eps=self.eps))
                else:
                    group_results[label] = dict(mean=float('nan'))
        self.reset()
        return total_results, group_results

    def reset(self):
        self.activations_pred, self.activations_target = [], []

    def _get_activations(self, batch):
        act = self.model(batch)
        act = act[0] if isinstance(act, (tuple, list)) else act
        return act.reshape(act.shape[0], -1)


# ----------------------------------------------------------------------------------------------------
# evaluator.py
# ----------------------------------------------------------------------------------------------------

def move_to_device(obj, device):
    """evaluation/utils.py:14-23."""
    if isinstance(obj, nn.Module) or torch.is_tensor(obj):
        return obj.to(device)
    if isinstance(obj, (tuple, list)):
        return [move_to_device(el, device) for el in obj]
    if isinstance(obj, dict):
        return {name: move_to_device(val, device) for name, val in obj.items()}
    raise ValueError(f'Unexpected type {type(obj)}')


def _interval_names(bins: int, bin_edges: np.ndarray):
    num_digits = max(0, math.ceil(math.log10(bins)) - 1)
    names = []
    for i in range(bins):
        a, b = round(100 * bin_edges[i], num_digits), round(100 * bin_edges[i + 1], num_digits)
        names.append('{0}-{1}%'.format('{:.{n}f}'.format(a, n=num_digits), '{:.{n}f}'.format(b, n=num_digits)))
    return names


class InpaintingEvaluator:
    """evaluator.py:16-111: scores over a dataset of dicts {image, mask[, inpainted]}, overall and grouped by masked-area share."""

    def __init__(self, dataset, scores, area_grouping=True, bins=10, batch_size=32, device='cuda', integral_func=None,
                 integral_title=None, clamp_image_range=None):
        self.scores, self.dataset = scores, dataset
        self.area_grouping, self.bins = area_grouping, bins
        self.device = torch.device(device)
        self.dataloader = torch.utils.data.DataLoader(self.dataset, shuffle=False, batch_size=batch_size)
        self.integral_func, self.integral_title, self.clamp_image_range = integral_func, integral_title, clamp_image_range

    def _get_bin_edges(self):
        bin_edges = np.linspace(0, 1, self.bins + 1)
        interval_names = _interval_names(self.bins, bin_edges)
        groups = []
        for batch in self.dataloader:
            mask = batch['mask']
            area = mask.to(self.device).reshape(mask.shape[0], -1).float().mean(dim=-1)
            bin_indices = np.searchsorted(bin_edges, area.detach().cpu().numpy(), side='right') - 1
            bin_indices[bin_indices == self.bins] = self.bins - 1          # area == 1 belongs to the last bin
            groups.append(bin_indices)
        return np.hstack(groups), interval_names

    def evaluate(self, model=None):
        results = {}
        groups, interval_names = self._get_bin_edges() if self.area_grouping else (None, None)
        for score_name, score in self.scores.items():
            score.to(self.device)
            with torch.no_grad():
                score.reset()
                for batch in self.dataloader:
                    batch = move_to_device(batch, self.device)
                    image_batch, mask_batch = batch['image'], batch['mask']
                    if self.clamp_image_range is not None:
                        image_batch = torch.clamp(image_batch, min=self.clamp_image_range[0], max=self.clamp_image_range[1])
                    if model is None:
                        assert 'inpainted' in batch, 'Model is None, so we expected precomputed inpainting results at key "inpainted"'
                        inpainted_batch = batch['inpainted']
                    else:
                        inpainted_batch = model(image_batch, mask_batch)
                    score(inpainted_batch, image_batch, mask_batch)
                total_results, group_results = score.get_value(groups=groups)
            results[(score_name, 'total')] = total_results
            if groups is not None:
                for group_index, group_values in group_results.items():
                    results[(score_name, interval_names[group_index])] = group_values
        if self.integral_func is not None:
            results[(self.integral_title, 'total')] = dict(mean=self.integral_func(results))
        return results


def ssim_fid100_f1(metrics, fid_scale=100):
    """evaluator.py:114-119."""
    ssim = metrics[('ssim', 'total')]['mean']
    fid = metrics[('fid', 'total')]['mean']
    fid_rel = max(0, fid_scale - fid) / fid_scale
    return 2 * ssim * fid_rel / (ssim + fid_rel + 1e-3)


def lpips_fid100_f1(metrics, fid_scale=100):
    """evaluator.py:122-127."""
    neg_lpips = 1 - metrics[('lpips', 'total')]['mean']
    fid = metrics[('fid', 'total')]['mean']
    fid_rel = max(0, fid_scale - fid) / fid_scale
    return 2 * neg_lpips * fid_rel / (neg_lpips + fid_rel + 1e-3)


class InpaintingEvaluatorOnline(nn.Module):
    """evaluator.py:131-220: per-batch accumulation (validation loops), finalised by ``evaluation_end``."""

    def __init__(self, scores, bins=10, image_key='image', inpainted_key='inpainted', integral_func=None, integral_title=None,
                 clamp_image_range=None):
        super().__init__()
        self.scores = nn.ModuleDict(scores)
        self.image_key, self.inpainted_key = image_key, inpainted_key
        self.bins_num = bins
        self.bin_edges = np.linspace(0, 1, self.bins_num + 1)
        self.interval_names = _interval_names(self.bins_num, self.bin_edges)
        self.groups = []
        self.integral_func, self.integral_title, self.clamp_image_range = integral_func, integral_title, clamp_image_range

    def _get_bins(self, mask_batch):
        area = mask_batch.reshape(mask_batch.shape[0], -1).float().mean(dim=-1).detach().cpu().numpy()
        return np.clip(np.searchsorted(self.bin_edges, area) - 1, 0, self.bins_num - 1)

    def forward(self, batch: Dict[str, torch.Tensor]):
        result = {}
        with torch.no_grad():
            image_batch, mask_batch, inpainted_batch = batch[self.image_key], batch['mask'], batch[self.inpainted_key]
            if self.clamp_image_range is not None:
                image_batch = torch.clamp(image_batch, min=self.clamp_image_range[0], max=self.clamp_image_range[1])
            self.groups.extend(self._get_bins(mask_batch))
            for score_name, score in self.scores.items():
                result[score_name] = score(inpainted_batch, image_batch, mask_batch)
        return result

    def process_batch(self, batch):
        return self(batch)

    def evaluation_end(self, states=None):
        self.groups = np.array(self.groups)
        results = {}
        for score_name, score in self.scores.items():
            cur_states = [s[score_name] for s in states] if states is not None else None
            total_results, group_results = score.get_value(groups=self.groups, states=cur_states)
            results[(score_name, 'total')] = total_results
            for group_index, group_values in group_results.items():
                results[(score_name, self.interval_names[group_index])] = group_values
        if self.integral_func is not None:
            results[(self.integral_title, 'total')] = dict(mean=self.integral_func(results))
        self.groups = []
        for sc in self.scores.values():
            sc.reset()
        return results


def make_evaluator(kind='default', ssim=True, lpips=False, fid=False, integral_kind=None, lpips_net=None, fid_net=None, **kwargs):
    """evaluation/__init__.py:9-33.  lpips / fid default to OFF here (they need downloaded networks: pass ``lpips_net`` / ``fid_net``)."""
    metrics = {}
    if ssim:
        metrics['ssim'] = SSIMScore()
    if lpips:
        metrics['lpips'] = LPIPSScore(net=lpips_net)
    if fid:
        metrics['fid'] = FIDScore(net=fid_net)
    integral_func = {None: None, 'ssim_fid100_f1': ssim_fid100_f1, 'lpips_fid100_f1': lpips_fid100_f1}.get(integral_kind, False)
    if integral_func is False:
        raise ValueError(f'Unexpected integral_kind={integral_kind}')
    if kind == 'default':
        return InpaintingEvaluatorOnline(scores=metrics, integral_func=integral_func, integral_title=integral_kind, **kwargs)
    raise ValueError(f'Unexpected evaluator kind={kind}')


# ----------------------------------------------------------------------------------------------------
# data.py:120-134 + bin/evaluate_predicts.py
# ----------------------------------------------------------------------------------------------------

class PrecomputedInpaintingResultsDataset(torch.utils.data.Dataset):
    """evaluation/data.py:58-83,120-134: ``datadir`` with ``<name>_maskNNN.png`` + ``<name><img_suffix>`` (predict.py's input contract),
    ``predictdir`` with ``<mask name><inpainted_suffix>``; items are CHW float32 in [0, 1]."""

    def __init__(self, datadir, predictdir, inpainted_suffix='_inpainted.jpg', img_suffix='.jpg', pad_out_to_modulo=None, scale_factor=None):
        from . import predict as P
        if scale_factor is not None:
            raise LamaError('dataset.scale_factor needs cv2.resize (not installed); not used by the shipped evaluation configs')
        self._P = P
        self.datadir = datadir
        pairs = P.list_dataset(datadir, img_suffix)                       # evaluation/data.py:59-62
        self.mask_filenames, self.img_filenames = [m for m, _ in pairs], [i for _, i in pairs]
        self.pad_out_to_modulo = pad_out_to_modulo
        if not datadir.endswith('/'):
            datadir += '/'
        self.pred_filenames = [os.path.join(predictdir, os.path.splitext(fname[len(datadir):])[0] + inpainted_suffix)
                               for fname in self.mask_filenames]

    def __len__(self):
        return len(self.mask_filenames)

    def __getitem__(self, i):
        P = self._P
        result = dict(image=P.load_image(self.img_filenames[i], mode='RGB'), mask=P.load_image(self.mask_filenames[i], mode='L')[None, ...],
                      inpainted=P.load_image(self.pred_filenames[i]))
        if self.pad_out_to_modulo is not None and self.pad_out_to_modulo > 1:
            for k in result:
                result[k] = P.pad_img_to_modulo(result[k], self.pad_out_to_modulo)
        return result


def main(argv=None) -> int:
    """bin/evaluate_predicts.py with the metrics available offline: ``python -m lama_amd.evaluation <config.yaml> <datadir> <predictdir>
    <outpath>``; the config's ``dataset_kwargs`` / ``evaluator_kwargs`` are honoured, the table is written tab-separated."""
    import argparse
    import yaml
    ap = argparse.ArgumentParser()
    ap.add_argument('config'), ap.add_argument('datadir'), ap.add_argument('predictdir'), ap.add_argument('outpath')
    args = ap.parse_args(argv)
    with open(args.config) as f:
        config = yaml.safe_load(f) or {}
    dataset = PrecomputedInpaintingResultsDataset(args.datadir, args.predictdir, **(config.get('dataset_kwargs') or {}))
    evaluator = InpaintingEvaluator(dataset, scores={'ssim': SSIMScore()}, **(config.get('evaluator_kwargs') or {}))
    results = evaluator.evaluate()
    os.makedirs(os.path.dirname(os.path.abspath(args.outpath)), exist_ok=True)
    import pandas as pd
    table = pd.DataFrame(results).stack(1).unstack(0)
    table.dropna(axis=1, how='all', inplace=True)
    table.to_csv(args.outpath, sep='\t', float_format='%.4f')
    print(table)
    return 0


if __name__ == '__main__':
    raise SystemExit(main())
