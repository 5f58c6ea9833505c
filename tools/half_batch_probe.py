#!/usr/bin/env python3
"""Experiment: one batch of 8 as TWO half-batches of 4 on two streams, the second started a fraction of a layer later, so that the
in-phase parts of one half (epilogues at the HBM rate, prologues, GEMM tails; every kernel of a half occupies only half of the CUs)
sit under the MFMA phases of the other.  Prints ms per 8 images for: one batch-8 graph; two batch-4 graphs launched on two streams (with
and without a start offset).  Result on MI355X / ROCm 7.2 (profiles/r02_half_batch_probe.txt): graphs launched on different streams do
NOT overlap at all (2 x the single-graph time), so the idea cannot be evaluated this way."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from lama_amd import _lib as L  # noqa: E402

dev = torch.device('cuda')
K = 20


def gen_of():
    m = bench.build_model(dev, L.PREC_F16X3)
    m.generator.use_graph = True
    return m.generator


g8, ga, gb = gen_of(), gen_of(), gen_of()
x8 = torch.rand(8, 4, 512, 512, device=dev)
xa, xb = x8[:4].contiguous(), x8[4:].contiguous()
for _ in range(3):
    y8 = g8(x8)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K):
    y8 = g8(x8)
torch.cuda.synchronize()
print(f'one batch-8 forward: {(time.perf_counter() - t0) / K * 1e3:.3f} ms per 8 images', flush=True)

sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
with torch.cuda.stream(sa):
    for _ in range(3):
        ya = ga(xa)
with torch.cuda.stream(sb):
    for _ in range(3):
        yb = gb(xb)
torch.cuda.synchronize()
assert torch.equal(torch.cat([ya, yb]), y8), 'half batches differ from the full batch'
pa, pb = next(iter(ga._plans.values())), next(iter(gb._plans.values()))


def run_pair(delay_cycles):
    # replay the captured graphs directly (no per-forward host sync): stream b starts `delay` later
    with torch.cuda.stream(sa):
        pa['graph'].replay()
    with torch.cuda.stream(sb):
        if delay_cycles:
            torch.cuda._sleep(delay_cycles)
        pb['graph'].replay()


for delay_us in (0, 100, 250):
    cyc = int(delay_us * 1e-6 * 2.1e9)
    for _ in range(3):
        run_pair(cyc)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        run_pair(cyc)
        sa.synchronize(); sb.synchronize()          # one "step" = both halves done
    dt = (time.perf_counter() - t0) / K * 1e3
    print(f'two batch-4 graphs on two streams, second delayed {delay_us:3d} us: {dt:.3f} ms per 8 images', flush=True)
# (Both halves as two branches of ONE captured graph: hipGraph instantiation segfaults in capture_end on ROCm 7.2 -- not pursued.)
with torch.cuda.stream(sa):
    for _ in range(K):
        pa['graph'].replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
with torch.cuda.stream(sa):
    for _ in range(K):
        pa['graph'].replay()
torch.cuda.synchronize()
print(f'one batch-4 graph alone: {(time.perf_counter() - t0) / K * 1e3:.3f} ms per 4 images', flush=True)
