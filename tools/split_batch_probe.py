#!/usr/bin/env python3
"""Does the chip do better on two half batches side by side than on one batch in lock-step?  (round 5 probe)

Every launch of the 8 x 512^2 forward fills the chip with workgroups that all run the same phase at the same time (load / MFMA / drain), and the
MFMA phases run at ~1.5 GHz because all 256 CUs draw matrix-core power at once.  Two half batches as two PARALLEL BRANCHES OF ONE hipGraph (kernel
branches of a graph do run concurrently on ROCm 7.2 -- round 2's spectral branch did; two separate graphs and graph + copies do not) would put
memory-bound launches of one half beside MFMA-bound launches of the other.  This tool measures it with the plans the generator builds for batch 4
(optionally with LAMA_TOOL_LIB=<profiling build> and LAMA_CW_G12=2 etc. to force the full-chip kernel geometries on the half launches).
usage: python tools/split_batch_probe.py [steps=30] [delay_cycles=0]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _toollib  # noqa: E402,F401
import bench  # noqa: E402
from lama_amd import _lib as L  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    delay = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    dev = torch.device('cuda', 0)
    model = bench.build_model(dev, L.PREC_F16X3)
    gen = model.generator
    gen.defer_range_check = True
    if os.environ.get('PROBE_FUSE1'):       # with LAMA_CW_G12=2 (profiling build): conv1 of the next layer rides in the half launches too
        from lama_amd import ffc as F
        F._DEFAULT_EXEC.fuse1_min_tiles = int(os.environ['PROBE_FUSE1'])
    img, mask = bench.synthetic_batch(dev, 1234)
    x = torch.cat([img * (1 - mask), mask], 1).contiguous()
    B = x.shape[0]

    def timeit(fn, n):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    gen.use_graph = True
    gen.clone_output = False
    ref = gen(x).clone()
    t_full = timeit(lambda: gen(x), steps)
    print(f'one batch-{B} graph: {t_full:.3f} ms per {B} images', flush=True)

    # PROBE_SPLIT part-batch plans with their own buffers, outputs side by side in one tensor
    gen.use_graph = False
    ns = int(os.environ.get('PROBE_SPLIT', '2'))
    h = B // ns
    xs = [x[i * h:(i + 1) * h] for i in range(ns)]
    plans = [gen._build_plan(xi.shape, dev) for xi in xs]
    out = torch.empty_like(ref)
    for i, pl in enumerate(plans):
        pl['bufs'][pl['out']] = out[i * h:(i + 1) * h]
    sides = [torch.cuda.Stream(device=dev) for _ in range(ns - 1)]

    def both(parallel):
        main_s = torch.cuda.current_stream(dev)
        if parallel:
            for i, s2 in enumerate(sides):
                s2.wait_stream(main_s)
                with torch.cuda.stream(s2):
                    if delay:
                        torch.cuda._sleep(delay * (i + 1))
                    gen._run_plan(plans[i + 1], xs[i + 1])
            gen._run_plan(plans[0], xs[0])
            for s2 in sides:
                main_s.wait_stream(s2)
        else:
            for pl, xi in zip(plans, xs):
                gen._run_plan(pl, xi)

    with gen._exec.range_scope(x, gen.precision, deferred=True):
        both(False)                                     # warm-up: packs, builds
        torch.cuda.synchronize()
        err = float((out - ref).abs().max())
        print(f'{ns} batch-{h} plans, eager: max |diff| vs the batch-{B} plan {err:.2e}', flush=True)
        for name, par in (('back to back in one graph', False), (f'as {ns} parallel branches of one graph', True)):
            g = torch.cuda.CUDAGraph()
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                both(par)
            torch.cuda.current_stream(dev).wait_stream(side)
            with torch.cuda.graph(g):
                both(par)
            out.zero_()
            t = timeit(g.replay, steps)
            err = float((out - ref).abs().max())
            print(f'{ns} batch-{h} plans {name}: {t:.3f} ms per {B} images ({t_full / t:.3f}x the batch-{B} graph), max |diff| {err:.2e}', flush=True)
    print('range ok:', gen.check_range(dev))


if __name__ == '__main__':
    main()
