#!/usr/bin/env python3
"""Target of a rocprofv3 --kernel-trace run (tools/session.sh statspy:tools/split_trace.py): N replays of the one-part graph, a marker, N replays of
the split graph (generator.split_batch = LAMA_TRACE_SPLIT, default the generator's rule) -- tools/overlap_summary.py then reads which kernels ran side
by side in each window."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _toollib  # noqa: E402,F401
import bench  # noqa: E402
from lama_amd import _lib as L  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    dev = torch.device('cuda', 0)
    model = bench.build_model(dev, L.PREC_F16X3)
    gen = model.generator
    gen.defer_range_check = True
    gen.use_graph = True
    gen.clone_output = False
    img, mask = bench.synthetic_batch(dev, 1234)
    x = torch.cat([img * (1 - mask), mask], 1).contiguous()
    for split in (1, int(os.environ.get('LAMA_TRACE_SPLIT', '0')) or None):
        gen.split_batch = split
        gen._plans.clear()
        for _ in range(3):
            gen(x)
        torch.cuda.synchronize()
        torch.cuda._sleep(20000)          # torch's spin_kernel alone on the GPU: the window separator
        torch.cuda.synchronize()
        for _ in range(n):
            gen(x)
        torch.cuda.synchronize()
        torch.cuda._sleep(20000)
        torch.cuda.synchronize()
    print('parts:', gen._split_parts(x.shape, dev), 'range ok:', gen.check_range(dev))


if __name__ == '__main__':
    main()
