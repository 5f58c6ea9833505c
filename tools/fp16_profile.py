#!/usr/bin/env python3
"""Graph replays of the generator at BASELINE configs[2] (4 x 1024^2, LAMA_PREC_F16) -- the target of a rocprofv3 --kernel-trace run."""
import os, sys
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
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    y = gen(x)
torch.cuda.synchronize()
print('done', float(y.mean()))
