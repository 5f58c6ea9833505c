#!/usr/bin/env python3
"""N hipGraph replays (and N eager runs) of the big-lama generator with the local convs as a chain of their own on the second stream
(ffc.SidePipe) must equal the one-stream run bit for bit, on alternating inputs.  Mismatches are counted on the device.
    python tools/pipeline_stress.py [N=400] [batch=8] [res=512]"""
import sys
import time
import torch
sys.path.insert(0, '.')
from lama_amd.modules import make_generator
from lama_amd import ffc as F
from oracle import lama_oracle as O      # synthetic weights / inputs only (test infrastructure)

N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
R = int(sys.argv[3]) if len(sys.argv) > 3 else 512
cfg = O.BIG_LAMA
gen = make_generator(None, kind='ffc_resnet', **cfg)
gen.load_state_dict(O.make_synthetic_state_dict(cfg, seed=0, calib_hw=64), strict=True)
gen.cuda()
xs = []
for seed in (12, 13):
    b = O.make_synthetic_batch(B, R, R, seed=seed)
    xs.append(torch.cat([b['image'] * (1 - b['mask']), b['mask']], 1).cuda())
gen.overlap_streams = False
F._DEFAULT_EXEC.cooperative_serial = True
refs = [gen(x).clone() for x in xs]
F._DEFAULT_EXEC.cooperative_serial = False
gen._plans.clear()
gen.overlap_streams = True
gen.pipeline_local = True
for graph in (True, False):
    gen.use_graph = graph
    gen._plans.clear()
    bad = torch.zeros(1, dtype=torch.int64, device='cuda')
    t0 = time.time()
    for it in range(N):
        bad += (~torch.eq(gen(xs[it & 1]), refs[it & 1]).all()).long()
    torch.cuda.synchronize()
    print(f'pipeline_local={gen.pipeline_local} graph={graph}: {N} runs, {int(bad)} mismatching, {time.time() - t0:.1f} s', flush=True)
