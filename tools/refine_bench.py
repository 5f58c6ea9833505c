#!/usr/bin/env python3
"""Wall time of refine_predict (BASELINE configs[4]: big-lama, one 2048 x 2048 image, refiner.px_budget=4194304, n_iters=15, max_scales=3)
on synthetic weights / input.  usage: refine_bench.py [res=2048] [n_iters=15]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from lama_amd import _lib as L  # noqa: E402
from lama_amd import refinement as R  # noqa: E402

res = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
n_iters = int(sys.argv[2]) if len(sys.argv) > 2 else 15
dev = torch.device('cuda')
model = bench.build_model(dev, L.PREC_F16X3)
g = torch.Generator().manual_seed(5)
img = torch.rand(1, 3, res, res, generator=g)
mask = torch.zeros(1, 1, res, res)
mask[:, :, res // 4: res // 2, res // 4: 3 * res // 4] = 1.0
for rep in range(2):
    batch = dict(image=img.to(dev), mask=mask.to(dev), unpad_to_size=[torch.tensor([res]), torch.tensor([res])])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = R.refine_predict(batch, model, gpu_ids='0,', modulo=8, n_iters=n_iters, lr=0.002, min_side=512, max_scales=3, px_budget=4194304)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f'refine_predict {res}x{res}, n_iters={n_iters}, 3 scales: {dt:.2f} s (run {rep}), out {tuple(out.shape)}, '
          f'peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB', flush=True)
