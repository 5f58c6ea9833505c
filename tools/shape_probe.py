#!/usr/bin/env python3
"""Step time of the hot path (mask compose -> generator -> blend -> u8) at arbitrary padded shapes: `shape_probe.py BxHxW [BxHxW ...]`.
Prints ms per step, ms per image and ns per pixel for every shape (graph replays; PROBE_STEPS, default 10) and, with PROBE_KERNELS=1,
the per-launch HIP-event times of one instrumented eager step.  Run it under `session.sh <tag> "statspy:tools/shape_probe.py ..."` for
rocprofv3 kernel stats of a shape.  Shapes must be multiples of 8 (what evaluation/data.py:29-33 pads to)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _toollib  # noqa: E402,F401
import bench  # noqa: E402
from lama_amd import _lib as L  # noqa: E402


def batch_of(b, h, w, device, seed=5):
    g = torch.Generator().manual_seed(seed)
    img = torch.floor(torch.rand(b, 3, h, w, generator=g) * 256).clamp_(0, 255) / 255.0
    mask = torch.zeros(b, 1, h, w)
    mask[:, :, h // 4: h // 4 + h // 2, w // 4: w // 4 + w // 2] = 1.0
    return img.to(device), mask.to(device)


def main():
    shapes = [tuple(int(v) for v in s.split('x')) for s in sys.argv[1:]] or [(1, 1080, 1920), (1, 1024, 2048)]
    steps = int(os.environ.get('PROBE_STEPS', '10'))
    device = torch.device('cuda', 0)
    lib = L.get_lib()
    timer = bench.KernelTimer(lib) if os.environ.get('PROBE_KERNELS') else None
    model = bench.build_model(device, L.PREC_NAMES[os.environ.get('PROBE_PREC', 'f16x3')])
    model.keep_predicted_image = False
    if 'PROBE_OVERLAP' in os.environ:          # 0: every layer's launches on ONE stream (no spectral branch beside the local conv)
        model.generator.overlap_streams = bool(int(os.environ['PROBE_OVERLAP']))
    st = torch.cuda.current_stream().cuda_stream
    for (b, h, w) in shapes:
        img, mask = batch_of(b, h, w, device)
        u8 = torch.empty(b, h, w, 3, dtype=torch.uint8, device=device)

        def step():
            o = model(dict(image=img, mask=mask))
            lib.quantize_u8_hwc(L.view(o['inpainted']), u8, b, h, w, st)

        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        print(f'{b}x{h}x{w} (planes {h // 8}x{w // 8}): {dt * 1e3:.3f} ms/step  {dt * 1e3 / b:.3f} ms/image  {dt * 1e9 / (b * h * w):.3f} ns/px', flush=True)
        if timer is not None:
            model.generator.use_graph = False
            model.generator._plans.clear()
            step()
            torch.cuda.synchronize()
            timer.records.clear()
            timer.on = True
            step()
            torch.cuda.synchronize()
            timer.on = False
            rows = sorted(timer.summary().items(), key=lambda kv: -kv[1]['total_us'])
            tot = sum(v['total_us'] for _, v in rows)
            for k, v in rows:
                print(f'    {k:72s} n={v["n"]:3d} avg {v["avg_us"]:9.1f} us  total {v["total_us"]:10.1f} us  {100 * v["total_us"] / tot:5.1f} %')
            model.generator.use_graph = True
        model.generator._plans.clear()
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
