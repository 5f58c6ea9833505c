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

    # two half-batch plans with their own buffers, outputs side by side in one tensor
    gen.use_graph = False
    h = B // 2
    xa, xb = x[:h], x[h:]
    pa, pb = gen._build_plan(xa.shape, dev), gen._build_plan(xb.shape, dev)
    out = torch.empty_like(ref)
    pa['bufs'][pa['out']], pb['bufs'][pb['out']] = out[:h], out[h:]
    s2 = torch.cuda.Stream(device=dev)

    def both(parallel):
        main_s = torch.cuda.current_stream(dev)
        if parallel:
            s2.wait_stream(main_s)
            with torch.cuda.stream(s2):
                if delay:
                    torch.cuda._sleep(delay)
                gen._run_plan(pb, xb)
            gen._run_plan(pa, xa)
            main_s.wait_stream(s2)
        else:
            gen._run_plan(pa, xa)
            gen._run_plan(pb, xb)

    with gen._exec.range_scope(x, gen.precision, deferred=True):
        both(False)                                     # warm-up: packs, builds
        torch.cuda.synchronize()
        err = float((out - ref).abs().max())
        print(f'two batch-{h} plans, eager: max |diff| vs the batch-{B} plan {err:.2e}', flush=True)
        for name, par in (('back to back in one graph', False), ('as two parallel branches of one graph', True)):
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
            print(f'two batch-{h} plans {name}: {t:.3f} ms per {B} images ({t_full / t:.3f}x the batch-{B} graph), max |diff| {err:.2e}', flush=True)
    print('range ok:', gen.check_range(dev))


if __name__ == '__main__':
    main()
