#!/usr/bin/env python3
"""Time graph replays of the generator at BASELINE configs[2] (4 x 1024^2) in LAMA_PREC_F16 -- for same-box A/B runs with the profiling
library's switches.   python tools/fp16_ab.py [replays=20]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from lama_amd import _lib as L  # noqa: E402
dev = torch.device('cuda')
model = bench.build_model(dev, L.PREC_F16)
gen = model.generator
gen.use_graph = True
x = torch.rand(4, 4, 1024, 1024, device=dev)
for _ in range(4):
    y = gen(x)
torch.cuda.synchronize()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
t0 = time.perf_counter()
for _ in range(n):
    y = gen(x)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f'{4 / dt:.2f} images/s {dt * 1e3:.3f} ms')
