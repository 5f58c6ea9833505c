#!/usr/bin/env python3
"""Batched, data-parallel mirror of ``bin/predict.py`` (reference lines 38-100) on the HIP path.

    python -m lama_amd.predict model.path=<dir> indir=<dir> outdir=<dir> \
        [model.checkpoint=best.ckpt] [dataset.img_suffix=.png] [dataset.pad_out_to_modulo=8] [out_ext=.png] \
        [batch_size=8] [precision=f16x3|bf16x3|f32] [io_threads=8]
    python -m torch.distributed.run --nproc-per-node N -m lama_amd.predict ...        # one process per GPU

Same on-disk contract as the reference: masks are ``**/*mask*.png`` (sorted, recursive), the image of a mask
is ``<mask path up to '_mask'><img_suffix>`` (``evaluation/data.py:59-62``), the result of a mask is written to
``<outdir>/<mask path relative to indir, extension replaced by out_ext>`` (``bin/predict.py:69-72``) as
``clip(inpainted*255, 0, 255).astype(uint8)`` cropped to the unpadded size (``bin/predict.py:86-94``).

What differs from the reference's batch-1 Python loop: images are padded to ``pad_out_to_modulo`` and *bucketed by
padded shape*, each bucket is cut into batches, batches are dealt round-robin to the ranks (one process per GPU,
weights replicated, no communication during compute), the u8 HWC results are produced on the device and the only
collective is one gather of those output images to rank 0 per round (RCCL over xGMI on GPUs, gloo in the CPU tests); rank 0
writes the PNGs on a thread pool so file IO overlaps the next round's compute.
"""
from __future__ import annotations

import glob
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib as L
from . import config as lcfg
from . import trainers

DEFAULTS = {'model.checkpoint': 'best.ckpt', 'dataset.img_suffix': '.png', 'dataset.pad_out_to_modulo': 8,
            'out_ext': '.png', 'out_key': 'inpainted', 'batch_size': 8, 'precision': 'f16x3',
            'io_threads': 8}   # configs/prediction/default.yaml (+ batch_size / precision / io_threads: this driver's own)


# ----------------------------------------------------------------------------------------------------------------
# dataset glue (saicinpainting/evaluation/data.py:12-33,58-83)
# ----------------------------------------------------------------------------------------------------------------

# keys of configs/prediction/default.yaml this driver honours; anything else is rejected instead of silently ignored
KNOWN_KEYS = set(DEFAULTS) | {'model.path', 'indir', 'outdir', 'device', 'dataset.kind', 'dataset.scale_factor', 'refine', 'profile',
                              'refiner.gpu_ids', 'refiner.modulo', 'refiner.n_iters', 'refiner.lr', 'refiner.min_side',
                              'refiner.max_scales', 'refiner.px_budget'}
REFINER_DEFAULTS = {'refiner.gpu_ids': '0,', 'refiner.modulo': 8, 'refiner.n_iters': 15, 'refiner.lr': 0.002, 'refiner.min_side': 512,
                    'refiner.max_scales': 3, 'refiner.px_budget': 1800000}     # configs/prediction/default.yaml:16-24


# option types: paths, suffixes and names stay strings whatever they look like (indir=2024, model.checkpoint=100 are a directory and
# a file name); everything else is a bool / int / float as in the YAML defaults
STRING_KEYS = {'model.path', 'model.checkpoint', 'indir', 'outdir', 'device', 'dataset.kind', 'dataset.img_suffix', 'out_ext', 'out_key',
               'precision', 'refiner.gpu_ids'}
BOOL_KEYS = {'refine', 'profile'}
INT_KEYS = {'dataset.pad_out_to_modulo', 'batch_size', 'io_threads', 'refiner.modulo', 'refiner.n_iters', 'refiner.min_side', 'refiner.max_scales',
            'refiner.px_budget'}
FLOAT_KEYS = {'refiner.lr', 'dataset.scale_factor'}


def _parse_value(k: str, v: str):
    if k in STRING_KEYS:
        return v
    try:
        if k in BOOL_KEYS:
            if v.lower() not in ('true', 'false'):
                raise ValueError(v)
            return v.lower() == 'true'
        if k in INT_KEYS:
            return int(v)
        if k in FLOAT_KEYS:
            return float(v)
    except ValueError:
        raise SystemExit(f'{k}={v!r}: expected a {"bool" if k in BOOL_KEYS else "number"}')
    raise AssertionError(f'option {k!r} has no declared type')


def parse_overrides(argv: Sequence[str]) -> Dict[str, object]:
    """Hydra-style ``a.b=c`` overrides on top of configs/prediction/default.yaml's defaults.  Unknown keys and options of the
    reference this driver does not implement are errors (the reference would act on them; ignoring them silently would not be
    a drop-in)."""
    cfg = dict(DEFAULTS)
    cfg.update(REFINER_DEFAULTS)
    for a in argv:
        if '=' not in a:
            raise SystemExit(f'expected key=value, got {a!r}')
        k, v = a.split('=', 1)
        if k not in KNOWN_KEYS:
            raise SystemExit(f'unknown option {k!r}; known: {sorted(KNOWN_KEYS)}')
        cfg[k] = _parse_value(k, v)
    for need in ('model.path', 'indir', 'outdir'):
        if need not in cfg:
            raise SystemExit(f'missing {need}=...')
    if str(cfg.get('device', 'cuda')).split(':')[0] != 'cuda':
        raise L.LamaError(f"device={cfg['device']}: lama_amd runs on an MI355X only (there is no CPU path)")
    if cfg.get('dataset.kind', 'default') != 'default':
        raise NotImplementedError(f"dataset.kind={cfg['dataset.kind']} (only the default InpaintingDataset of bin/predict.py)")
    if cfg.get('out_key', 'inpainted') not in ('inpainted', 'predicted_image'):
        raise SystemExit(f"out_key={cfg['out_key']!r}: the predict_only forward (trainers/default.py:56-71) produces 'inpainted' and 'predicted_image'")
    if cfg.get('dataset.scale_factor') is not None and not cfg['dataset.scale_factor'] > 0:
        raise SystemExit(f"dataset.scale_factor={cfg['dataset.scale_factor']!r}: a positive factor (evaluation/data.py:74-77)")
    return cfg


def list_dataset(indir: str, img_suffix: str = '.png') -> List[Tuple[str, str]]:
    """(mask path, image path) pairs: InpaintingDataset.__init__, evaluation/data.py:59-62."""
    masks = sorted(glob.glob(os.path.join(indir, '**', '*mask*.png'), recursive=True))
    return [(m, m.rsplit('_mask', 1)[0] + img_suffix) for m in masks]


def load_image(fname: str, mode: str = 'RGB') -> np.ndarray:
    """evaluation/data.py:12-20 (PIL instead of cv2: same decoded values)."""
    from PIL import Image
    img = np.array(Image.open(fname).convert(mode))
    if img.ndim == 3:
        img = np.transpose(img, (2, 0, 1))
    return img.astype('float32') / 255


def ceil_modulo(x: int, mod: int) -> int:
    return x if x % mod == 0 else (x // mod + 1) * mod


def pad_img_to_modulo(img: np.ndarray, mod: int) -> np.ndarray:
    """evaluation/data.py:29-33: symmetric padding at the bottom / right."""
    c, h, w = img.shape
    return np.pad(img, ((0, 0), (0, ceil_modulo(h, mod) - h), (0, ceil_modulo(w, mod) - w)), mode='symmetric')


# ---- dataset.scale_factor (evaluation/data.py:42-55,74-77): cv2.resize(img, dsize=None, fx=f, fy=f) with INTER_AREA on the float image and
# INTER_NEAREST on the float mask.  cv2 is not in this image, so this is a restatement of OpenCV 4's imgproc/resize.cpp from its published
# algorithm -- PARITY UNPINNED (no cv2 here to generate vectors from): the index / weight tables and the float32 accumulation order follow
# resize.cpp (computeResizeAreaTab + ResizeArea_Invoker; resizeAreaFast_Invoker's four-at-a-time sums; the INTER_AREA-upscaling coefficients of
# the linear path; resizeNN's floor(x / f) taps), known-answer tests in tests/test_scale_factor.py.  Host code on purpose: it is the data loader.

def scaled_size(h: int, w: int, factor: float) -> Tuple[int, int]:
    """dsize of cv::resize(dsize=None, fx, fy): saturate_cast<int>(size * f) = round half to even."""
    return int(round(h * factor)), int(round(w * factor))


def _area_taps(ssize: int, dsize: int, scale: float):
    """computeResizeAreaTab: per destination index the source taps and their float32 weights, padded to a common tap count with weight 0."""
    import math
    taps = []
    for d in range(dsize):
        f1 = d * scale
        f2 = f1 + scale
        cell = min(scale, ssize - f1)
        s1, s2 = math.ceil(f1), math.floor(f2)
        s2 = min(s2, ssize - 1)
        s1 = min(s1, s2)
        t = []
        if s1 - f1 > 1e-3:
            t.append((s1 - 1, np.float32((s1 - f1) / cell)))
        for sx in range(s1, s2):
            t.append((sx, np.float32(1.0 / cell)))
        if f2 - s2 > 1e-3:
            t.append((s2, np.float32(min(min(f2 - s2, 1.0), cell) / cell)))
        taps.append(t)
    T = max(len(t) for t in taps)
    idx = np.zeros((dsize, T), np.int64)
    wt = np.zeros((dsize, T), np.float32)
    for d, t in enumerate(taps):
        for k, (sx, a) in enumerate(t):
            idx[d, k], wt[d, k] = sx, a
    return idx, wt


def _resize_area(img: np.ndarray, factor: float) -> np.ndarray:
    """INTER_AREA of a float32 [H, W] or [H, W, C] image by ``factor`` in both directions."""
    import math
    a = np.ascontiguousarray(img, dtype=np.float32)
    H, W = a.shape[:2]
    dh, dw = scaled_size(H, W, factor)
    if dh <= 0 or dw <= 0:
        raise L.LamaError(f'scale_factor {factor}: a {W}x{H} image scales to nothing')
    scale = 1.0 / factor
    if scale >= 1.0:                                            # shrinking: true area averaging
        isc = int(round(scale))
        if abs(scale - isc) < np.finfo(np.float64).eps:         # integer factor: resizeAreaFast -- box sums in groups of four, * 1 / area
            # destination pixels whose box would leave the image are cut off by dsize = round(size * f) (<= size / isc only when it divides)
            dh2, dw2 = min(dh, H // isc), min(dw, W // isc)
            box = [a[sy:sy + dh2 * isc:isc, sx:sx + dw2 * isc:isc] for sy in range(isc) for sx in range(isc)]
            acc = np.zeros_like(box[0])
            k, area = 0, isc * isc
            while k <= area - 4:
                acc = acc + (((box[k] + box[k + 1]) + box[k + 2]) + box[k + 3])
                k += 4
            while k < area:
                acc = acc + box[k]
                k += 1
            out = acc * np.float32(1.0 / area)
            if (dh2, dw2) != (dh, dw):                          # (x.5 rounded up: the partial last boxes average the pixels inside the image)
                full = np.zeros((dh, dw) + a.shape[2:], np.float32)
                full[:dh2, :dw2] = out
                for yy in range(dh):
                    for xx in range(dw):
                        if yy >= dh2 or xx >= dw2:
                            blk = a[yy * isc:min((yy + 1) * isc, H), xx * isc:min((xx + 1) * isc, W)]
                            px = blk.reshape(-1, *a.shape[2:])                   # resizeAreaFast's border loop: sum / count of the pixels inside
                            acc1 = np.zeros(a.shape[2:], np.float32)
                            for v in px:
                                acc1 = acc1 + v
                            full[yy, xx] = acc1 / np.float32(len(px))
                out = full
            return out
        xi, xw = _area_taps(W, dw, scale)
        yi, yw = _area_taps(H, dh, scale)
        ex = (slice(None),) * 2 + (None,) * (a.ndim - 2)
        buf = np.zeros((H, dw) + a.shape[2:], np.float32)      # the x pass of every source row: buf[dx] += S[sx] * alpha, taps in table order
        for t in range(xi.shape[1]):
            buf = buf + a[:, xi[:, t]] * xw[None, :, t][ex] if t else a[:, xi[:, t]] * xw[None, :, t][ex]
        out = np.zeros((dh, dw) + a.shape[2:], np.float32)     # the y pass: sum[dx] = beta * buf, then += for the following rows
        for t in range(yi.shape[1]):
            term = buf[yi[:, t]] * yw[:, t][(slice(None), None) + (None,) * (a.ndim - 2)]
            out = out + term if t else term
        return out
    # enlarging with INTER_AREA = the linear path with area coefficients: sx = floor(dx * scale), fx = (dx + 1) - (sx + 1) * f, kept in [0, 1)
    def coef(ssize, dsize):
        sx = np.floor(np.arange(dsize) * scale).astype(np.int64)
        fx = ((np.arange(dsize) + 1) - (sx + 1) * factor).astype(np.float32)
        fx = np.where(fx <= 0, np.float32(0), fx - np.floor(fx)).astype(np.float32)
        lo = sx < 0
        fx[lo], sx[lo] = 0, 0
        hi = sx >= ssize - 1
        fx[hi], sx[hi] = 0, ssize - 1
        return sx, np.minimum(sx + 1, ssize - 1), (np.float32(1) - fx).astype(np.float32), fx
    x0, x1, ax0, ax1 = coef(W, dw)
    y0, y1, ay0, ay1 = coef(H, dh)
    ex = (None, slice(None)) + (None,) * (a.ndim - 2)
    rows = a[:, x0] * ax0[ex] + a[:, x1] * ax1[ex]
    ey = (slice(None), None) + (None,) * (a.ndim - 2)
    return (rows[y0] * ay0[ey] + rows[y1] * ay1[ey]).astype(np.float32)


def _resize_nearest(img: np.ndarray, factor: float) -> np.ndarray:
    """INTER_NEAREST: tap min(floor(d / f), size - 1) in each direction (resizeNN)."""
    H, W = img.shape[:2]
    dh, dw = scaled_size(H, W, factor)
    if dh <= 0 or dw <= 0:
        raise L.LamaError(f'scale_factor {factor}: a {W}x{H} image scales to nothing')
    inv = 1.0 / factor
    ys = np.minimum(np.floor(np.arange(dh) * inv).astype(np.int64), H - 1)
    xs = np.minimum(np.floor(np.arange(dw) * inv).astype(np.int64), W - 1)
    return np.ascontiguousarray(img[ys][:, xs])


def scale_image(img: np.ndarray, factor: float, interpolation: str = 'area') -> np.ndarray:
    """evaluation/data.py:42-55: a float32 [C,H,W] image through cv2.resize(fx=fy=factor) and back to [C,H',W'] ('area' = cv2.INTER_AREA, the
    default; 'nearest' = cv2.INTER_NEAREST, what InpaintingDataset uses for the mask)."""
    hwc = img[0] if img.shape[0] == 1 else np.transpose(img, (1, 2, 0))
    out = _resize_nearest(hwc, factor) if interpolation == 'nearest' else _resize_area(hwc, factor)
    return out[None, ...] if out.ndim == 2 else np.transpose(out, (2, 0, 1))


def load_item_u8(mask_path: str, img_path: str):
    """The same pair as it is on disk: (u8 image [H,W,3], u8 mask [H,W], (H, W)).  ``load_image``'s / 255, ``pad_img_to_modulo`` and
    bin/predict.py:84's ``mask > 0`` happen on the device (lama_mask_compose_u8_fwd, ABI v110): the host only decodes."""
    from PIL import Image
    image = np.asarray(Image.open(img_path).convert('RGB'))
    mask = np.asarray(Image.open(mask_path).convert('L'))
    if tuple(mask.shape) != tuple(image.shape[:2]):
        raise L.LamaError(f'{mask_path}: mask is {mask.shape[1]}x{mask.shape[0]} but {img_path} is {image.shape[1]}x{image.shape[0]}')
    return image, mask, tuple(image.shape[:2])


def load_item(mask_path: str, img_path: str, pad_mod: int, scale_factor: Optional[float] = None):
    """InpaintingDataset.__getitem__, evaluation/data.py:69-83 -> (image [3,H',W'], mask [1,H',W'], (H, W)): ``scale_factor`` rescales the image
    (INTER_AREA) and the mask (INTER_NEAREST) before the padding, and (H, W) = ``unpad_to_size`` is the rescaled size (data.py:74-79)."""
    image = load_image(img_path, 'RGB')
    mask = load_image(mask_path, 'L')[None, ...]
    if tuple(mask.shape[1:]) != tuple(image.shape[1:]):
        raise L.LamaError(f'{mask_path}: mask is {mask.shape[2]}x{mask.shape[1]} but {img_path} is {image.shape[2]}x{image.shape[1]}')
    if scale_factor is not None:
        image = scale_image(image, scale_factor)
        mask = scale_image(mask, scale_factor, interpolation='nearest')
    hw = image.shape[1:]
    if pad_mod and pad_mod > 1:
        image, mask = pad_img_to_modulo(image, pad_mod), pad_img_to_modulo(mask, pad_mod)
    return image, mask, hw


# ----------------------------------------------------------------------------------------------------------------
# work distribution
# ----------------------------------------------------------------------------------------------------------------

def plan_rounds(shapes: Sequence[Tuple[int, int]], batch_size: int, world: int) -> List[dict]:
    """Bucket item indices by padded shape, cut buckets into batches, deal the batches of a bucket round-robin to ranks.
    Returns rounds: dict(shape=(H', W'), batches=[item-index list per rank (possibly empty)])."""
    buckets: Dict[Tuple[int, int], List[int]] = {}
    for i, s in enumerate(shapes):
        buckets.setdefault(tuple(s), []).append(i)
    rounds = []
    # largest padded shape first: the activation / pinned buffers of every later bucket then fit blocks the caching allocators already hold
    # (ascending order paid a round of hipMalloc / hipHostMalloc calls per new shape: ~30 ms each in a directory of differently sized photos)
    for shape in sorted(buckets, key=lambda hw: (-hw[0] * hw[1], hw)):
        idx = buckets[shape]
        batches = [idx[i:i + batch_size] for i in range(0, len(idx), batch_size)]
        for r0 in range(0, len(batches), world):
            chunk = batches[r0:r0 + world]
            rounds.append(dict(shape=shape, batches=chunk + [[] for _ in range(world - len(chunk))]))
    return rounds


def gather_to_root(dist, gathered: Optional[torch.Tensor], part: torch.Tensor, rank: int, world: int, root: int = 0):
    """The data path's only collective: the u8 result images of every rank to the WRITER rank.  ``dist.gather`` (RCCL: one
    send per non-root rank, ``world - 1`` receives on the root, point-to-point over xGMI) -- the other ranks allocate and receive nothing,
    unlike an all-gather, which moves ``world`` times the bytes to deliver copies nobody reads.  ``gathered`` = [world * n, ...] on the root,
    None elsewhere.  Returns the asynchronous work handle."""
    parts = list(gathered.chunk(world, dim=0)) if rank == root else None          # contiguous slices of the root's buffer: no staging copy
    return dist.gather(part, gather_list=parts, dst=root, async_op=True)


PROFILE_SPANS = False                    # profile=true: HostFedStep records timing events around every step's compute
LOOP_TIMES: Dict[str, float] = {}      # wall seconds of the main thread per phase of the round loop (python -m lama_amd.predict ... profile=true prints them)
TUNE_MIN_ROUNDS = 1024    # buckets with fewer rounds skip the split-plan timing check: ~0.35 s for at most 5 % of the bucket's time (HostFedStep(tune=False))
CAPTURE_MIN_ROUNDS = 32   # ... and with fewer than this also the graph capture: ~30 ms for ~8 % of a round's 10 ms (HostFedStep(capture=False): plain launches)


class _RangeRestart(Exception):
    """A forward of this run left the fp16 split's range: the generator is on the bf16 split now, the run starts over."""


def encode_png(rgb: np.ndarray) -> bytes:
    """An 8-bit RGB (or gray) image as PNG bytes with the parameters cv2.imwrite uses by default (bin/predict.py:94: IMWRITE_PNG_COMPRESSION
    unset -> libpng filter SUB, zlib level 1, strategy Z_RLE): the Sub filter as one numpy subtraction, one zlib stream, one IDAT chunk.
    Same decoded pixels as any PNG writer; 7.4 ms per 512 x 512 image where PIL's encoder (adaptive filter choice per row, default strategy)
    takes 56 ms at the same level -- the CLI was encode-bound on the host at 0.47 of the GPU rate (bench.py predict_cli_leg, round 6)."""
    import struct
    import zlib
    a = np.ascontiguousarray(rgb)
    if a.dtype != np.uint8 or a.ndim not in (2, 3) or (a.ndim == 3 and a.shape[2] not in (1, 3)):
        raise L.LamaError(f'encode_png: uint8 [H,W] / [H,W,3] image, got {a.dtype} {a.shape}')
    h, w = a.shape[:2]
    c = 1 if a.ndim == 2 else a.shape[2]
    flat = a.reshape(h, w * c)
    raw = np.empty((h, 1 + w * c), np.uint8)
    raw[:, 0] = 1                                              # filter type Sub: byte - byte of the pixel to the left (mod 256)
    raw[:, 1:1 + c] = flat[:, :c]
    raw[:, 1 + c:] = flat[:, c:] - flat[:, :-c]
    co = zlib.compressobj(1, zlib.DEFLATED, 15, 8, zlib.Z_RLE)
    data = co.compress(raw.tobytes()) + co.flush()

    def chunk(tag: bytes, body: bytes) -> bytes:
        return struct.pack('>I', len(body)) + tag + body + struct.pack('>I', zlib.crc32(tag + body) & 0xffffffff)

    ihdr = struct.pack('>IIBBBBB', w, h, 8, 2 if c == 3 else 0, 0, 0, 0)
    return b'\x89PNG\r\n\x1a\n' + chunk(b'IHDR', ihdr) + chunk(b'IDAT', data) + chunk(b'IEND', b'')


class _Latch:
    """Count-down latch: the round loop waits on it before a pinned result set is overwritten while pool workers still snapshot it."""

    def __init__(self):
        import threading
        self._c, self._n = threading.Condition(), 0

    def add(self, n: int = 1):
        with self._c:
            self._n += n

    def done(self):
        with self._c:
            self._n -= 1
            if self._n <= 0:
                self._c.notify_all()

    def wait(self):
        with self._c:
            while self._n > 0:
                self._c.wait()


def _snap_and_write(path: str, view: np.ndarray, latch: _Latch, pool):
    """``fast`` pool job: copy the image out of the pinned result set (then release the latch: the set may be overwritten) and queue its encode +
    write on ``pool``; returns that future."""
    try:
        rgb = view.copy()
    finally:
        latch.done()
    return pool.submit(_write_png, path, rgb)


def _write_png(path: str, rgb: np.ndarray):
    """bin/predict.py:93-94 (the reference converts RGB -> BGR only because cv2.imwrite expects BGR).  ``.png``: encode_png; any other
    extension goes through PIL, which picks the format from it as cv2.imwrite does."""
    os.makedirs(os.path.dirname(path), exist_ok=True)
    if path.lower().endswith('.png'):
        with open(path, 'wb') as f:
            f.write(encode_png(rgb))
        return
    from PIL import Image
    Image.fromarray(rgb).save(path)


# ----------------------------------------------------------------------------------------------------------------
# one step fed from host memory
# ----------------------------------------------------------------------------------------------------------------

class HostFedStep:
    """One predict step -- mask binarisation (bin/predict.py:84), mask-compose + generator + blend (trainers/default.py:56-71), u8 HWC
    quantisation (bin/predict.py:92) -- fed from / drained to pinned HOST buffers, double-buffered:

        launch(p):   H2D  host input set 1-p -> device set 1-p        (the NEXT batch; copy stream)
                  || compute on device set p -> u8[p]                 (THIS batch)
                  || D2H  u8[1-p] -> host result set 1-p              (the PREVIOUS batch; copy stream; ``drain``)

    ``mode='replay'``: the copies run on streams of their own, forked and joined by events, beside the replay of the generator's own
    hipGraph (+ four plain launches: mask compose, blend, u8).  ``'streams'``: the same around ~270 plain launches.  ``'graph'``: the three as
    parallel branches INSIDE one captured hipGraph per parity (built in round 5).  Measured (bench.py value_host_fed, images/s; the boxes of
    the pool differ in GPU AND host speed; profiles/r05_host_fed.txt):

        box (resident)      serial (copies on the compute stream)   replay    streams    graph
        r4 driver (825)     761                                     788       819        --
        r05a      (800)     740                                     --        765        736
        r05b      (817)     754                                     --        740        756
        r05h      (852)     760                                     790       531        788     (the batch in four parallel parts)
        r05i      (837)     708                                     757       530        777     (the batch in four parallel parts)

    A replay of a graph overlaps copies of OTHER streams only partly (the r4 result); copy nodes INSIDE a graph run in line with its kernel
    nodes when the graph is one kernel chain (r05a / r05b: = the serial form) but overlap when it has parallel kernel branches already
    (r05h / r05i: the split plans of generator.split_batch -- the executor then runs the graph on several queues); plain launches overlap
    fully but make the step launch-bound on a slow host (r05b) and with the batch in four parts.  ``mode='auto'`` (the default): 'graph'
    where the generator splits this shape, else 'replay' -- neither ever lost to the serial form.

    The caller alternates p = 0, 1, 0, ...: it fills host set 1-p before ``launch(p)`` (after ``wait`` has told it that the launch that last
    read that set is complete), calls ``prime(p0)`` once before the first launch and ``flush(p_last)`` after the last.  ``drain=False``
    (multi-rank: the results go through the gather instead) leaves the D2H branch out.  On a CPU device (the emulator tests) the same body
    runs synchronously, unpinned."""

    def __init__(self, model, batch_size: int, Hp: int, Wp: int, device, *, drain: bool = True, binarize: bool = True, mode: str = 'auto',
                 u8_input: bool = True, tune: bool = True, out_key: str = 'inpainted', capture: bool = True):
        self.model, self.n, self.Hp, self.Wp = model, int(batch_size), int(Hp), int(Wp)
        if out_key not in ('inpainted', 'predicted_image'):                        # bin/predict.py:86: batch[predict_config.out_key]
            raise L.LamaError(f"out_key {out_key!r}: 'inpainted' or 'predicted_image'")
        self.out_key = out_key
        self.device = torch.device(device)
        self.on_gpu = self.device.type == 'cuda'
        self.drain, self.binarize = drain, binarize
        if mode not in ('auto', 'replay', 'streams', 'graph', 'host'):
            raise L.LamaError(f'HostFedStep mode {mode!r}: auto, replay, streams, graph or host')
        self.one_part = False
        if mode == 'auto' and not capture:
            # a bucket of a few rounds (round 6, third session): not even ONE graph capture -- the one-part plan as plain launches beside
            # the copy streams (a warm-up + capture + instantiate is 10-40 ms per new shape, tools/newshape_probe.py; the ~270 launches of a step
            # cost the host 3 ms, less than the step takes on the GPU at any size)
            mode, self.one_part = 'streams', True
        elif mode == 'auto' and not tune:
            # a bucket of tens to hundreds of rounds: the one-part plan, captured once, in the host-synchronised form -- tune_split's two plans, two
            # captures and 18 replays (~0.35 s) buy at most 5 % of the bucket's time
            mode, self.one_part = ('host' if self.on_gpu else 'replay'), True
        if mode == 'auto':      # by measurement (table above): copy nodes overlap only in a graph that has parallel kernel branches already
            gen = model.generator
            split = 1
            if hasattr(gen, '_split_parts'):
                # (the generator verifies once per shape that this runtime runs the parts side by side: FFCResNetGenerator.verify_split)
                keep_ag, gen._assume_graph = getattr(gen, '_assume_graph', False), True
                try:
                    split = gen.tune_split((self.n, 4, Hp, Wp), self.device) if self.on_gpu else 1
                finally:
                    gen._assume_graph = keep_ag
            mode = 'host' if self.on_gpu else 'replay'       # round 6 (u8 copies): the host-synchronised copy streams beside whichever plan won, 0.999 of the resident rate (profiles/r06_host_fed.txt)
        self.mode = mode
        pin = dict(pin_memory=True) if self.on_gpu else {}
        # round 6: the step is fed with what is on disk -- u8 HWC image, u8 mask, (h, w) per image: 4 bytes per pixel over PCIe instead of 16, no
        # float conversion / padding on the host (forward_u8).  ``u8_input=False`` keeps the fp32 NCHW form of round 5 (A/B runs, callers with float data).
        self.u8_input = bool(u8_input) and hasattr(model, 'forward_u8')
        if self.u8_input:
            self.h_img = [torch.zeros(self.n, Hp, Wp, 3, dtype=torch.uint8, **pin) for _ in range(2)]
            self.h_mask = [torch.zeros(self.n, Hp, Wp, dtype=torch.uint8, **pin) for _ in range(2)]
            self.h_sizes = [torch.zeros(self.n, 2, dtype=torch.int32, **pin) for _ in range(2)]
            self.d_img = [torch.zeros(self.n, Hp, Wp, 3, dtype=torch.uint8, device=self.device) for _ in range(2)]
            self.d_mask = [torch.zeros(self.n, Hp, Wp, dtype=torch.uint8, device=self.device) for _ in range(2)]
            self.d_sizes = [torch.zeros(self.n, 2, dtype=torch.int32, device=self.device) for _ in range(2)]
        else:
            self.h_img = [torch.zeros(self.n, 3, Hp, Wp, dtype=torch.float32, **pin) for _ in range(2)]
            self.h_mask = [torch.zeros(self.n, 1, Hp, Wp, dtype=torch.float32, **pin) for _ in range(2)]
            self.d_img = [torch.zeros(self.n, 3, Hp, Wp, dtype=torch.float32, device=self.device) for _ in range(2)]
            self.d_mask = [torch.zeros(self.n, 1, Hp, Wp, dtype=torch.float32, device=self.device) for _ in range(2)]
            self.h_sizes = self.d_sizes = None
        self.h_u8 = [torch.zeros(self.n, Hp, Wp, 3, dtype=torch.uint8, **pin) for _ in range(2)] if drain else None
        self.u8 = [torch.zeros(self.n, Hp, Wp, 3, dtype=torch.uint8, device=self.device) for _ in range(2)]
        self.graphs = [None, None]
        self.span_events = None       # a list: launch() (mode 'host') appends (start, end) timing events around each step's compute
        self.done = [torch.cuda.Event() for _ in range(2)] if self.on_gpu else None
        self._launched = [False, False]
        self._h2d_issued = [False, False]
        if self.on_gpu:
            self.h2d = [torch.cuda.Event() for _ in range(2)]
            self.d2h = [torch.cuda.Event() for _ in range(2)]
            self.s_in, self.s_out = torch.cuda.Stream(device=self.device), torch.cuda.Stream(device=self.device)

    # -- host views -----------------------------------------------------------------------------------------------
    def host(self, p: int):
        """The pinned host set p as numpy views: (image, mask) -- u8 [n,Hp,Wp,3] / [n,Hp,Wp] (``u8_input``; ``put`` fills them) or fp32 NCHW."""
        return self.h_img[p].numpy(), self.h_mask[p].numpy()

    def put(self, p: int, j: int, image: Optional[np.ndarray], mask: Optional[np.ndarray]):
        """``u8_input``: image j of host set p = a decoded u8 HWC image [h,w,3] + u8 mask [h,w] (top-left corner of the slot), or None: an empty
        slot of a partial batch."""
        if not self.u8_input:
            # fp32 NCHW form (dataset.scale_factor: the rescaled image is not u8 any more): image [3,Hp,Wp] / mask [1,Hp,Wp] float32 as
            # ``load_item`` returns them, already padded; an empty slot is zeros
            hi, hm = self.h_img[p].numpy(), self.h_mask[p].numpy()
            if image is None:
                hi[j] = 0
                hm[j] = 0
                return
            if tuple(image.shape) != (3, self.Hp, self.Wp) or tuple(mask.shape) != (1, self.Hp, self.Wp):
                raise L.LamaError(f'image {tuple(image.shape)} / mask {tuple(mask.shape)}: padded [3,{self.Hp},{self.Wp}] / [1,{self.Hp},{self.Wp}] float32')
            hi[j] = image
            hm[j] = mask
            return
        sz = self.h_sizes[p].numpy()
        if image is None:
            sz[j] = (0, 0)
            return
        h, w = image.shape[:2]
        if h > self.Hp or w > self.Wp:
            raise L.LamaError(f'image {h}x{w} does not fit the bucket {self.Hp}x{self.Wp}')
        self.h_img[p].numpy()[j, :h, :w] = image
        self.h_mask[p].numpy()[j, :h, :w] = mask
        sz[j] = (h, w)

    def result(self, p: int) -> np.ndarray:
        return self.h_u8[p].numpy()

    # -- pieces ---------------------------------------------------------------------------------------------------
    def _compute(self, p: int):
        lib = self.model.generator._exec.lib
        if self.u8_input:                                                                          # bin/predict.py:82-92 in two elementwise launches + the generator
            with torch.no_grad():
                self.model.forward_u8(self.d_img[p], self.d_mask[p], self.d_sizes[p], self.u8[p], binarize=self.binarize, out_key=self.out_key)
            return
        mask = self.d_mask[p]
        batch = dict(image=self.d_img[p], mask=(mask > 0) * 1 if self.binarize else mask)          # bin/predict.py:84
        keep = self.model.keep_predicted_image
        self.model.keep_predicted_image = False
        try:
            with torch.no_grad():
                out = self.model(batch)[self.out_key]                                            # bin/predict.py:85-86
        finally:
            self.model.keep_predicted_image = keep
        stream = torch.cuda.current_stream(self.device).cuda_stream if self.on_gpu else 0
        lib.quantize_u8_hwc(L.view(out), self.u8[p], self.n, self.Hp, self.Wp, stream)             # bin/predict.py:92 on the device

    def _body(self, p: int):
        q = 1 - p
        if not self.on_gpu:
            self.d_img[q].copy_(self.h_img[q]); self.d_mask[q].copy_(self.h_mask[q])
            if self.u8_input:
                self.d_sizes[q].copy_(self.h_sizes[q])
            if self.drain:
                self.h_u8[q].copy_(self.u8[q])
            self._compute(p)
            return
        main = torch.cuda.current_stream(self.device)
        self.s_in.wait_stream(main)
        with torch.cuda.stream(self.s_in):
            self.d_img[q].copy_(self.h_img[q], non_blocking=True)
            self.d_mask[q].copy_(self.h_mask[q], non_blocking=True)
            if self.u8_input:
                self.d_sizes[q].copy_(self.h_sizes[q], non_blocking=True)
        if self.drain:
            self.s_out.wait_stream(main)
            with torch.cuda.stream(self.s_out):
                self.h_u8[q].copy_(self.u8[q], non_blocking=True)
        self._compute(p)
        main.wait_stream(self.s_in)
        if self.drain:
            main.wait_stream(self.s_out)

    def _capture(self, p: int):
        gen = self.model.generator
        keep = (gen.use_graph, gen.defer_range_check, getattr(gen, '_assume_graph', False))
        gen.use_graph, gen.defer_range_check = False, True       # the plan's launches go into THIS graph; no flag read-back (a host sync) inside it
        gen._assume_graph = True                                 # ... so the plan may be the split one although generator.use_graph is off
        try:
            if self.graphs[1 - p] is None:                       # first capture: pack the weights / build the plan outside it
                side = torch.cuda.Stream(device=self.device)
                side.wait_stream(torch.cuda.current_stream(self.device))
                with torch.cuda.stream(side):
                    self._compute(p)
                torch.cuda.current_stream(self.device).wait_stream(side)
            g = torch.cuda.CUDAGraph()
            # thread_local: RCCL's watchdog thread polls events of the gather that may still be in flight on the side stream
            with torch.cuda.graph(g, capture_error_mode='thread_local'):
                self._body(p)
            self.graphs[p] = g
        finally:
            gen.use_graph, gen.defer_range_check, gen._assume_graph = keep

    # -- the caller's four verbs ----------------------------------------------------------------------------------
    def prime(self, p: int):
        """H2D of host set p on the current stream (before the first launch of a run) and a host wait for it: the caller refills that host
        set for the batch after next before any ``wait`` has covered this copy."""
        self.d_img[p].copy_(self.h_img[p], non_blocking=self.on_gpu)
        self.d_mask[p].copy_(self.h_mask[p], non_blocking=self.on_gpu)
        if self.u8_input:
            self.d_sizes[p].copy_(self.h_sizes[p], non_blocking=self.on_gpu)
        if self.on_gpu:
            torch.cuda.current_stream(self.device).synchronize()

    def launch(self, p: int):
        if not self.on_gpu:
            gen = self.model.generator
            keep, gen.defer_range_check = gen.defer_range_check, True
            try:
                self._body(p)
            finally:
                gen.defer_range_check = keep
            return
        if self.mode == 'host':
            # round 6: NO device-side fork / join between the queues.  The compute of this batch is queued first (behind nothing but the event of
            # its own upload, issued a step ago and long complete); then the HOST waits for the previous step -- the last reader of device set
            # 1 - p and the producer of u8[1 - p] -- and only then queues the next upload and the previous download on the copy streams, which
            # therefore need no barrier against the compute queue.  With 4 bytes per pixel up and 3 down the copies are ~0.3 ms of a 9.7 ms step:
            # the cross-queue dependencies of the other forms cost more than that (profiles/r06_host_fed.txt).
            q = 1 - p
            main = torch.cuda.current_stream(self.device)
            gen = self.model.generator
            keep = (gen.use_graph, gen.defer_range_check, getattr(gen, 'split_batch', None))
            gen.use_graph, gen.defer_range_check = True, True
            if self.one_part:
                gen.split_batch = 1
            try:
                if self._h2d_issued[p]:
                    main.wait_event(self.h2d[p])
                if self.span_events is not None:       # profile=true: the step's span on the GPU clock
                    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                    ev[0].record(main)
                self._compute(p)
                if self.span_events is not None:
                    ev[1].record(main)
                    self.span_events.append(ev)
            finally:
                gen.use_graph, gen.defer_range_check = keep[:2]
                if self.one_part:
                    gen.split_batch = keep[2]
            self.done[p].record(main)
            if self._launched[q]:
                self.done[q].synchronize()               # host: step k - 1 is complete (step k is queued behind it: the GPU does not idle)
            with torch.cuda.stream(self.s_in):
                self.d_img[q].copy_(self.h_img[q], non_blocking=True)
                self.d_mask[q].copy_(self.h_mask[q], non_blocking=True)
                if self.u8_input:
                    self.d_sizes[q].copy_(self.h_sizes[q], non_blocking=True)
                self.h2d[q].record(self.s_in)
            self._h2d_issued[q] = True
            if self.drain:
                with torch.cuda.stream(self.s_out):
                    self.h_u8[q].copy_(self.u8[q], non_blocking=True)
                    self.d2h[q].record(self.s_out)
            self._launched[p] = True
            return
        if self.mode == 'graph':
            if self.graphs[p] is None:
                self._capture(p)
            self.graphs[p].replay()
        else:       # the two copies on streams of their own (fork / join by events) beside this batch's compute: the generator's own hipGraph
            gen = self.model.generator          # replay ('replay') or its ~270 plain launches ('streams')
            keep = (gen.use_graph, gen.defer_range_check, getattr(gen, 'split_batch', None))
            gen.use_graph, gen.defer_range_check = self.mode == 'replay', True
            if self.one_part:
                gen.split_batch = 1
            try:
                self._body(p)
            finally:
                gen.use_graph, gen.defer_range_check = keep[:2]
                if self.one_part:
                    gen.split_batch = keep[2]
        self.done[p].record(torch.cuda.current_stream(self.device))
        self._launched[p] = True

    def wait(self, p: int):
        """Host: until the most recent launch(p) is complete (its H2D has read host set 1-p, its D2H has filled result(1-p))."""
        if self.on_gpu and self._launched[p]:
            self.done[p].synchronize()
            if self.mode == 'host':                      # ... and the copies that launch queued on the copy streams
                self.h2d[1 - p].synchronize()
                if self.drain:
                    self.d2h[1 - p].synchronize()

    def flush(self, p: int):
        """D2H of u8[p] (the last launch's own result: nothing follows to carry it) and a host wait for it."""
        if self.drain:
            self.h_u8[p].copy_(self.u8[p], non_blocking=self.on_gpu)
        if self.on_gpu:
            torch.cuda.current_stream(self.device).synchronize()
            if self.mode == 'host':
                self.s_in.synchronize()
                self.s_out.synchronize()


# ----------------------------------------------------------------------------------------------------------------
# the predict loop
# ----------------------------------------------------------------------------------------------------------------

def predict(model: trainers.DefaultInpaintingTrainingModule, items: List[Tuple[str, str]], indir: str, outdir: str, *,
            pad_mod: int = 8, batch_size: int = 8, out_ext: str = '.png', device='cuda', rank: int = 0, world: int = 1,
            dist=None, io_threads: int = 8, out_key: str = 'inpainted', scale_factor: Optional[float] = None) -> int:
    """Run every (mask, image) pair through ``model`` and write the results (rank 0).  Returns the number of images written.
    ``scale_factor`` = ``dataset.scale_factor`` (evaluation/data.py:74-77): every image / mask is rescaled on the host before the padding, the
    step is then fed with fp32 tensors and the results have the rescaled size.

    Host pipeline: the PNGs of round r + 1 are decoded / padded on the thread pool while round r computes, results are
    written on the same pool; a partial last batch of a bucket is zero-padded to ``batch_size`` so that it replays the bucket's
    captured plan instead of building (and capturing) a second one, and a bucket's plan is dropped when the bucket is done."""
    gen = model.generator
    keep = gen.defer_range_check
    # no host synchronisation per forward: the fp16 split's range flag is read once per bucket, where every rank synchronises anyway
    gen.defer_range_check = True
    try:
        for _attempt in range(2):
            try:
                return _predict_once(model, items, indir, outdir, pad_mod=pad_mod, batch_size=batch_size, out_ext=out_ext, device=device,
                                     rank=rank, world=world, dist=dist, io_threads=io_threads, out_key=out_key, scale_factor=scale_factor)
            except _RangeRestart:
                continue                     # the generator switched to the 3-term bf16 split (every rank alike): all images again
        raise L.LamaError('range restart did not converge')
    finally:
        gen.defer_range_check = keep


def _predict_once(model, items, indir, outdir, *, pad_mod, batch_size, out_ext, device, rank, world, dist, io_threads, out_key='inpainted', scale_factor=None) -> int:
    from PIL import Image

    def header_size(item):                   # padded shape from the IMAGE's PNG header only: every rank builds the same plan
        with Image.open(item[1]) as im:
            w, h = im.size
        return (h, w) if scale_factor is None else scaled_size(h, w, scale_factor)

    lib = model.generator._exec.lib
    pool = ThreadPoolExecutor(io_threads)
    # the copies into / out of the pinned sets are on the round loop's critical path (the next launch waits for them) and take 0.1-0.3 ms each: they
    # get workers of their own instead of queueing behind milliseconds of PNG work in ``pool``
    fast = ThreadPoolExecutor(min(4, max(1, io_threads)))
    try:
        sizes = list(pool.map(header_size, items, chunksize=64))       # (0.15 ms per file on the main thread was a tenth of a long directory's wall time)
        shapes = [(ceil_modulo(h, pad_mod), ceil_modulo(w, pad_mod)) if pad_mod and pad_mod > 1 else (h, w) for h, w in sizes]
        rounds = plan_rounds(shapes, batch_size, world)
        return _predict_rounds(model, items, indir, outdir, rounds, sizes, pool, lib, pad_mod=pad_mod, batch_size=batch_size, out_ext=out_ext,
                               device=device, rank=rank, world=world, dist=dist, out_key=out_key, fast=fast, scale_factor=scale_factor)
    finally:
        fast.shutdown()
        pool.shutdown()                      # on every exit path (a raise out of a bucket included): no worker thread outlives the call


def _predict_rounds(model, items, indir, outdir, rounds, sizes, pool, lib, *, pad_mod, batch_size, out_ext, device, rank, world, dist, out_key='inpainted', fast=None,
                    scale_factor=None) -> int:
    """The rounds of one attempt.  Per bucket (padded shape) a ``HostFedStep``: round k computes from device input set k & 1 while the SAME graph
    launch uploads round k + 1's decoded images into the other set and (one rank) downloads round k - 1's u8 results; the host fills the
    pinned set of round k + 1 and queues round k - 2's PNG writes while round k runs.  Several ranks: the results go through the one
    collective -- ``gather_to_root`` of the u8 batch, on a side stream behind the step -- and rank 0 downloads the gathered rounds."""
    futures, written = [], 0                 # futures of the snapshot jobs; each returns the future of its encode + write
    fast = fast if fast is not None else pool
    bucket_paths: List[str] = []             # the PNGs queued for the bucket whose range flag has not been read yet

    def drain():
        for f in futures:
            f.result().result()

    on_gpu = torch.device(device).type == 'cuda'
    side = torch.cuda.Stream(device=device) if on_gpu else None     # gather + D2H of the gathered rounds, beside the next round's compute

    loads: Dict[int, list] = {}

    def submit_loads(r):
        if r < len(rounds) and r not in loads:
            if scale_factor is None:
                loads[r] = [pool.submit(load_item_u8, *items[i]) for i in rounds[r]['batches'][rank]]  # decode only: / 255, padding, mask > 0 run on the device
            else:                                                                                      # data.py:69-83 on the host: decode, / 255, rescale, pad
                loads[r] = [pool.submit(load_item, *items[i], pad_mod, scale_factor) for i in rounds[r]['batches'][rank]]

    snapped = [_Latch(), _Latch()]           # per pinned result set: the workers that still copy their image out of it

    def write_round(rd, host, pset):
        """Rank 0: queue the PNG writes of one round from its u8 results on the host (``host`` = [ranks * batch_size, Hp, Wp, 3] = pinned result
        set ``pset``).  The workers snapshot their image themselves (the main thread only queues); ``snapped[pset].wait()`` before that set is
        downloaded into again."""
        nonlocal written
        for r, idxs in enumerate(rd['batches']):
            for j, i in enumerate(idxs):
                mask_path = items[i][0]
                h, w = sizes[i]
                rel = os.path.splitext(mask_path[len(indir):].lstrip(os.sep))[0] + out_ext      # bin/predict.py:69-72
                bucket_paths.append(os.path.join(outdir, rel))
                snapped[pset].add()
                futures.append(fast.submit(_snap_and_write, bucket_paths[-1], host[r * batch_size + j, :h, :w], snapped[pset], pool))
                written += 1

    submit_loads(0)
    submit_loads(1)
    r0 = 0
    while r0 < len(rounds):
        Hp, Wp = rounds[r0]['shape']
        r1 = r0
        while r1 < len(rounds) and rounds[r1]['shape'] == (Hp, Wp):
            r1 += 1
        K = r1 - r0
        t_ = time.perf_counter()
        hs = HostFedStep(model, batch_size, Hp, Wp, device, drain=(world == 1), tune=(K >= TUNE_MIN_ROUNDS), capture=(K >= CAPTURE_MIN_ROUNDS), out_key=out_key,
                         u8_input=(scale_factor is None))
        LOOP_TIMES['bucket_setup'] = LOOP_TIMES.get('bucket_setup', 0.0) + time.perf_counter() - t_
        if PROFILE_SPANS:
            hs.span_events = []
        gathered = h_out = None
        if world > 1 and rank == 0:
            gathered = [torch.empty(world * batch_size, Hp, Wp, 3, dtype=torch.uint8, device=device) for _ in range(2)]
            h_out = [torch.zeros(world * batch_size, Hp, Wp, 3, dtype=torch.uint8, **(dict(pin_memory=True) if on_gpu else {})) for _ in range(2)]
        works = [None, None]
        collected = [torch.cuda.Event() for _ in range(2)] if (on_gpu and world > 1) else None

        def fill(pp, r):
            t_ = time.perf_counter()
            loaded = [f.result() for f in loads.pop(r)]
            LOOP_TIMES['wait_decode'] = LOOP_TIMES.get('wait_decode', 0.0) + time.perf_counter() - t_
            t_ = time.perf_counter()

            def put(j):                                                               # partial (or, for a rank without a batch, empty) round: empty slots
                if j < len(loaded):
                    hs.put(pp, j, loaded[j][0], loaded[j][1])
                else:
                    hs.put(pp, j, None, None)
            list(fast.map(put, range(batch_size)))                                    # the copies into the pinned set, side by side (distinct slots)
            LOOP_TIMES['fill_pinned'] = LOOP_TIMES.get('fill_pinned', 0.0) + time.perf_counter() - t_

        def collect(pp):
            """Several ranks: the round that launch(pp) just computed goes through the gather (side stream, behind the step); rank 0 downloads it."""
            snapped[pp].wait()                    # h_out[pp] is downloaded into again: round k - 2's images have been copied out of it
            if on_gpu:
                with torch.cuda.stream(side):
                    side.wait_event(hs.done[pp])
                    works[pp] = gather_to_root(dist, gathered[pp] if rank == 0 else None, hs.u8[pp], rank, world)
                    works[pp].wait()                                                  # the side stream waits for RCCL's stream
                    if rank == 0:
                        h_out[pp].copy_(gathered[pp], non_blocking=True)
                    collected[pp].record(side)
            else:
                works[pp] = gather_to_root(dist, gathered[pp] if rank == 0 else None, hs.u8[pp], rank, world)
                works[pp].wait()
                if rank == 0:
                    h_out[pp].copy_(gathered[pp])

        fill(0, r0)
        submit_loads(r0 + 2)
        t_ = time.perf_counter()
        hs.prime(0)
        LOOP_TIMES['bucket_setup'] += time.perf_counter() - t_
        for k in range(K):
            pp = k & 1
            if k + 1 < K:
                fill(1 - pp, r0 + k + 1)          # (the launch that last read this host set -- step k - 2 -- is complete: waited for below, at k - 1)
            submit_loads(r0 + k + 2)
            submit_loads(r0 + k + 3)
            if on_gpu and works[pp] is not None:
                torch.cuda.current_stream(hs.device).wait_event(collected[pp])        # step k - 2's gather has read u8[pp]
            t_ = time.perf_counter()
            snapped[1 - pp].wait()                # (one rank) this launch downloads into result set 1 - pp: round k - 3's images have been copied out of it
            hs.launch(pp)
            LOOP_TIMES['launch'] = LOOP_TIMES.get('launch', 0.0) + time.perf_counter() - t_
            if world > 1:
                collect(pp)
                if k >= 1:                        # round k - 1: gathered and downloaded while step k runs
                    if on_gpu:
                        collected[1 - pp].synchronize()
                    if rank == 0:
                        write_round(rounds[r0 + k - 1], h_out[1 - pp].numpy(), 1 - pp)
            elif k >= 1:
                t_ = time.perf_counter()
                hs.wait(1 - pp)                   # step k - 1 is complete: it downloaded round k - 2 and read the host set filled above
                LOOP_TIMES['wait_step'] = LOOP_TIMES.get('wait_step', 0.0) + time.perf_counter() - t_
                if k >= 2:
                    t_ = time.perf_counter()
                    write_round(rounds[r0 + k - 2], hs.result(pp), pp)
                    LOOP_TIMES['queue_writes'] = LOOP_TIMES.get('queue_writes', 0.0) + time.perf_counter() - t_
        pl = (K - 1) & 1
        if world > 1:
            if on_gpu:
                collected[pl].synchronize()
            if rank == 0:
                write_round(rounds[r1 - 1], h_out[pl].numpy(), pl)
        else:
            hs.wait(pl)
            if K >= 2:
                write_round(rounds[r1 - 2], hs.result(1 - pl), 1 - pl)
            snapped[pl].wait()
            hs.flush(pl)
            write_round(rounds[r1 - 1], hs.result(pl), pl)
        if on_gpu:
            torch.cuda.current_stream(hs.device).synchronize()                      # every rank: the graphs / buffers below are idle now
            if hs.span_events:
                ev = hs.span_events
                LOOP_TIMES['gpu_step_spans'] = LOOP_TIMES.get('gpu_step_spans', 0.0) + sum(a.elapsed_time(b) for a, b in ev) * 1e-3
                LOOP_TIMES['gpu_first_start_to_last_end'] = LOOP_TIMES.get('gpu_first_start_to_last_end', 0.0) + ev[0][0].elapsed_time(ev[-1][1]) * 1e-3
                LOOP_TIMES['steps'] = LOOP_TIMES.get('steps', 0.0) + len(ev)

        # the bucket's ONE read-back of the fp16 split's range flag (the forwards above did not synchronise); all ranks decide alike
        def red(bad, _dev=hs.device):
            t = torch.tensor([1 if bad else 0], dtype=torch.int32, device=_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return int(t.item()) != 0
        # The PNG writes of this bucket are already queued (they overlap the compute); the flag is read only now.  Out-of-range forwards
        # produced garbage images: with auto_fallback the restart overwrites every file, WITHOUT it check_range raises -- and the bucket's
        # files are removed first, so that a LamaRangeError never leaves bad output on disk.
        try:
            in_range = model.generator.check_range(hs.device, reduce=red if world > 1 else None)
        except L.LamaRangeError:
            drain()
            for pth in bucket_paths:
                if os.path.exists(pth):
                    os.remove(pth)
            raise
        if not in_range:
            drain()
            raise _RangeRestart()
        bucket_paths.clear()
        model.generator.drop_plan((batch_size, 4, Hp, Wp), hs.device)               # bucket done: free its buffers / graphs
        del hs, gathered, h_out
        r0 = r1
    drain()
    return written


def predict_refine(model, items: List[Tuple[str, str]], indir: str, outdir: str, cfg: dict, *, device='cuda', rank: int = 0,
                   world: int = 1) -> int:
    """``refine=True`` (bin/predict.py:75-81): one image at a time through lama_amd.refinement.refine_predict.  Each image's
    optimisation is sequential, so the ranks are plain replicas: rank r takes items r, r + world, ... and writes its own results
    (no collective on this path)."""
    from .refinement import refine_predict
    pad_mod = int(cfg['dataset.pad_out_to_modulo'])
    pool = ThreadPoolExecutor(4)
    futures, written = [], 0
    for i in range(rank, len(items), world):
        mask_path, img_path = items[i]
        image, mask, (h, w) = load_item(mask_path, img_path, pad_mod, cfg.get('dataset.scale_factor'))
        batch = dict(image=torch.from_numpy(image)[None].to(device), mask=torch.from_numpy(mask)[None].to(device),
                     unpad_to_size=[torch.tensor([h]), torch.tensor([w])])
        # bin/predict.py:75-81: the refine branch hands the RAW grayscale mask to refine_predict (line 84's binarisation belongs to the
        # plain branch only); the refiner thresholds it itself at every scale (refinement.py:212,304-305)
        cur_res = refine_predict(batch, model, gpu_ids=str(cfg['refiner.gpu_ids']), modulo=int(cfg['refiner.modulo']),
                                 n_iters=int(cfg['refiner.n_iters']), lr=float(cfg['refiner.lr']), min_side=int(cfg['refiner.min_side']),
                                 max_scales=int(cfg['refiner.max_scales']), px_budget=int(cfg['refiner.px_budget']))
        res = cur_res[0].permute(1, 2, 0).numpy()                                                          # bin/predict.py:80
        u8 = np.clip(res * 255, 0, 255).astype('uint8')                                                    # bin/predict.py:92
        rel = os.path.splitext(mask_path[len(indir):].lstrip(os.sep))[0] + cfg['out_ext']
        futures.append(pool.submit(_write_png, os.path.join(outdir, rel), u8))
        written += 1
    for f in futures:
        f.result()
    pool.shutdown()
    return written


def main(argv: Optional[Sequence[str]] = None) -> int:
    cfg = parse_overrides(sys.argv[1:] if argv is None else argv)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise L.LamaError('lama_amd.predict needs an MI355X: no GPU is visible (there is no CPU fallback)')
    device = torch.device('cuda', local)
    torch.cuda.set_device(device)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', device_id=device)
    train_config = lcfg.load_train_config(os.path.join(cfg['model.path'], 'config.yaml'))           # bin/predict.py:46-48
    ckpt = os.path.join(cfg['model.path'], 'models', cfg['model.checkpoint'])                       # bin/predict.py:55-57
    model = trainers.load_checkpoint(train_config, ckpt, strict=False, map_location=device)         # bin/predict.py:58
    model.freeze()
    model.generator.set_precision(L.PREC_NAMES[cfg['precision']])
    model.generator.use_graph = True
    indir = cfg['indir'] if cfg['indir'].endswith(os.sep) else cfg['indir'] + os.sep               # bin/predict.py:63-64
    items = list_dataset(indir, cfg['dataset.img_suffix'])
    if cfg.get('refine', False):                                                                    # bin/predict.py:75-81
        n = predict_refine(model, items, indir, cfg['outdir'], cfg, device=device, rank=rank, world=world)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        print(f'rank {rank}: wrote {n} refined images to {cfg["outdir"]}')
        return 0
    if cfg.get('profile', False):
        global PROFILE_SPANS
        PROFILE_SPANS = True
    t0 = time.perf_counter()
    n = predict(model, items, indir, cfg['outdir'], pad_mod=int(cfg['dataset.pad_out_to_modulo']), batch_size=int(cfg['batch_size']),
                out_ext=cfg['out_ext'], device=device, rank=rank, world=world, dist=dist, io_threads=int(cfg['io_threads']), out_key=cfg['out_key'],
                scale_factor=cfg.get('dataset.scale_factor'))
    dt = time.perf_counter() - t0
    if rank == 0:
        # (the loop's own wall time: plan build + graph capture of every shape bucket, PNG decode, upload, compute, download, PNG encode + write)
        print(f'wrote {n} images to {cfg["outdir"]} in {dt:.3f} s ({n / max(dt, 1e-9):.1f} images/s, {world} rank(s), io_threads={int(cfg["io_threads"])})')
        if cfg.get('profile', False):
            print('main-thread seconds per phase of the round loop: ' + ', '.join(f'{k} {v:.3f}' for k, v in sorted(LOOP_TIMES.items())))
    if world > 1:
        dist.destroy_process_group()
    return 0


if __name__ == '__main__':
    sys.exit(main())
