#!/usr/bin/env python3
"""What a NEW padded shape costs the predict loop (batch 1): HostFedStep construction (pinned + device sets), first launch (plan build, warm-up,
graph capture in 'replay' mode), second launch, drop_plan -- per phase, for a few shapes.  usage: newshape_probe.py [mode=replay|streams]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from lama_amd import _lib as L  # noqa: E402
from lama_amd.predict import HostFedStep  # noqa: E402


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else 'auto'
    dev = torch.device('cuda', 0)
    model = bench.build_model(dev, L.PREC_F16X3)
    gen = model.generator
    gen.use_graph = True
    gen.defer_range_check = True
    sync = torch.cuda.synchronize
    for (H, W) in ((512, 512), (400, 600), (440, 656), (704, 1024), (960, 1400), (400, 600)):
        t = [time.perf_counter()]
        hs = HostFedStep(model, 1, H, W, dev, tune=False, mode=mode) if mode != 'auto' else HostFedStep(model, 1, H, W, dev, tune=False)
        sync(); t.append(time.perf_counter())
        hs.prime(0)
        sync(); t.append(time.perf_counter())
        hs.launch(0)
        sync(); t.append(time.perf_counter())
        hs.launch(1)
        sync(); t.append(time.perf_counter())
        hs.launch(0)
        sync(); t.append(time.perf_counter())
        hs.flush(0)
        gen.check_range(dev)
        gen.drop_plan((1, 4, H, W), dev)
        del hs
        sync(); t.append(time.perf_counter())
        d = [(b - a) * 1e3 for a, b in zip(t, t[1:])]
        print(f'{H}x{W} mode={mode}: construct {d[0]:.1f} ms, prime {d[1]:.1f}, launch#1 {d[2]:.1f}, launch#2 {d[3]:.1f}, launch#3 {d[4]:.1f}, flush+drop {d[5]:.1f}', flush=True)


if __name__ == '__main__':
    main()
