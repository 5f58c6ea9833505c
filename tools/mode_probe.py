#!/usr/bin/env python3
"""Is the one-part plan's 4-7 % run-to-run spread a property of WHERE its buffers sit?  Several fresh one-part and four-part plans in
ONE process, each on newly allocated buffers (an odd-sized spacer allocation in between shifts the addresses), 10 graph replays each.
usage: python tools/mode_probe.py [rounds=4]"""
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
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    dev = torch.device('cuda', 0)
    model = bench.build_model(dev, L.PREC_F16X3)
    gen = model.generator
    gen.defer_range_check = True
    gen.use_graph = True
    gen.clone_output = False
    gen.verify_split = False
    img, mask = bench.synthetic_batch(dev, 1234)
    x = torch.cat([img * (1 - mask), mask], 1).contiguous()
    spacers = []
    for r in range(rounds):
        for split in (1, 4):
            gen.split_batch = split
            gen._plans.clear()
            torch.cuda.empty_cache()
            for _ in range(3):
                gen(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                gen(x)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 10 * 1e3
            plan = next(iter(gen._plans.values()))
            p0 = (plan['parts'][0] if 'parts' in plan else plan)
            addr = p0['bufs'][sorted(p0['bufs'])[0]].data_ptr()
            print(f'round {r} parts {split}: {ms:.3f} ms per batch   (first buffer at 0x{addr:x})', flush=True)
        spacers.append(torch.empty((3 * r + 1) * 1234567, device=dev, dtype=torch.uint8))      # shifts the next round's allocations
    print('range ok:', gen.check_range(dev))


if __name__ == '__main__':
    main()
