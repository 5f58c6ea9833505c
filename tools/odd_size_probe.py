#!/usr/bin/env python3
"""Time the generator on a non-power-of-two input (bottleneck planes h/8 x w/8 take the generic DFT kernels).  usage: odd_size_probe.py H W"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from lama_amd import _lib as L  # noqa: E402
H, W = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device('cuda')
model = bench.build_model(dev, L.PREC_F16X3)
gen = model.generator
gen.use_graph = True
x = torch.rand(1, 4, H, W, device=dev)
for _ in range(2):
    y = gen(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 5
for _ in range(n):
    y = gen(x)
torch.cuda.synchronize()
print(f'{H}x{W} (planes {H // 8}x{W // 8}): {(time.perf_counter() - t0) / n * 1e3:.2f} ms per image', flush=True)
