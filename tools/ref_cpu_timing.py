#!/usr/bin/env python3
"""Build container only (needs /root/reference): the REFERENCE's own FFCResNetGenerator classes (imported with the two stubs of
tests/golden/make_golden.py) timed beside the oracle restatement on the same host, same weights, same input -- the evidence that
bench.py's `cpu_baseline` (kind "port": the restatement, because /root/reference does not exist on the GPU box) times the same
arithmetic at the same speed.  Modes of BASELINE.md section 3: batch-1 loop, one batched forward, one thread.
usage: python tools/ref_cpu_timing.py [threads=8] > profiles/r02_reference_cpu_timing.txt"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
from oracle import lama_oracle as O  # noqa: E402
import make_golden as MG  # noqa: E402

threads = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = O.BIG_LAMA
sd = O.make_synthetic_state_dict(cfg, seed=0, calib_hw=64)
gen = MG.ref_generator(cfg, sd)
batch = O.make_synthetic_batch(4, 512, 512, seed=1234)
x = torch.cat([batch['image'] * (1 - batch['mask']), batch['mask']], 1)


def timeit(fn, reps):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps


print(f'host: {os.cpu_count()} logical cores; torch {torch.__version__}; big-lama (synthetic calibrated weights), 512x512')
with torch.no_grad():
    d = float((gen(x[:1]) - O.generator_forward(x[:1], sd, cfg)).abs().max())
    print(f'max |reference classes - oracle restatement| on one image: {d:.2e}')
    for nt in (threads, 1):
        torch.set_num_threads(nt)
        reps = 3 if nt > 1 else 1
        tr = timeit(lambda: gen(x[:1]), reps)
        to = timeit(lambda: O.generator_forward(x[:1], sd, cfg), reps)
        print(f'threads={nt:3d}  batch-1 forward: reference classes {1 / tr:.3f} images/s, oracle restatement {1 / to:.3f} images/s (ratio {tr / to:.3f})')
    torch.set_num_threads(threads)
    tr = timeit(lambda: gen(x), 2)
    to = timeit(lambda: O.generator_forward(x, sd, cfg), 2)
    print(f'threads={threads:3d}  batch-4 forward: reference classes {4 / tr:.3f} images/s, oracle restatement {4 / to:.3f} images/s (ratio {tr / to:.3f})')
