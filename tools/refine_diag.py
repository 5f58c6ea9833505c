#!/usr/bin/env python3
"""GPU-box diagnostic (tests/tools only -- imports the oracle): one scale of refinement on the FULL big-lama generator, product
(HIP, explicit reverse pass) against the CPU oracle (torch autograd + Adam) ITERATION BY ITERATION: loss, gradient of z, z after
the step, final prediction.  usage: refine_diag.py [res=1024] [n_iters=4] [f32|default]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lama_amd import _lib as L, refinement as RF, trainers  # noqa: E402
from lama_amd.backward import RearPass  # noqa: E402
from oracle import lama_oracle as O, refine_oracle as R  # noqa: E402
import importlib.util  # noqa: E402

res = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n_iters = int(sys.argv[2]) if len(sys.argv) > 2 else 4
mode = sys.argv[3] if len(sys.argv) > 3 else 'default'
spec = importlib.util.spec_from_file_location('mk', os.path.join(ROOT, 'tests', 'golden', 'make_golden_refine.py'))
mk = importlib.util.module_from_spec(spec); spec.loader.exec_module(mk)
cfg = dict(O.BIG_LAMA)
sd = O.make_synthetic_state_dict(cfg, seed=0, calib_hw=64)
model = trainers.DefaultInpaintingTrainingModule(dict(generator=dict(kind='ffc_resnet', **cfg)))
model.load_state_dict({'generator.' + k: v for k, v in sd.items()}, strict=True)
model.freeze().cuda()
gen = model.generator
if mode == 'f32':
    gen.set_precision(L.PREC_F32)
image, mask = mk.make_case(res)
fri = R.first_resblock_index(cfg)
# lower-resolution reference: the plain forward at half resolution (same tensor for both sides)
half_img, half_msk = R.pyrdown(image), R.pyrdown_mask(mask)
with torch.no_grad():
    ref_low = model(dict(image=half_img.cuda(), mask=half_msk.cuda()))['inpainted'].cpu()
tr_ref, tr = dict(keep_z=True), dict(keep_z=True)
t0 = time.time()
out_ref = R.infer(image, mask, sd, cfg, ref_low, (res, res), n_iters=n_iters, lr=0.002, trace=tr_ref)
print(f'oracle: {time.time() - t0:.1f} s', flush=True)
rear = RearPass(gen, fri, bwd_precision=L.PREC_F32 if mode == 'f32' else L.PREC_BF16X3)
out = RF._infer(image.cuda(), mask.cuda(), gen.model[0:fri], [rear], ref_low.cuda(), (res, res), ['cuda'], 1, n_iters=n_iters, lr=0.002, trace=tr)
print('loss ref', np.array(tr_ref['loss'])); print('loss got', np.array(tr['loss']))
print('pred0 diff', float((tr['pred0'].cpu() - tr_ref['pred0']).abs().max()))
for k in range(len(tr['z'])):
    zr, gr = tr_ref['z'][k], tr_ref['g'][k]
    z, g = tr['z'][k].cpu(), tr['g'][k].cpu()
    dz, dg = (z - zr).abs(), (g - gr).abs()
    a = gr.abs()
    print(f'it {k}: |dz| mean {dz.mean():.2e} max {dz.max():.2e} | |dg| max {dg.max():.2e} rel-L2 {float(dg.norm() / gr.norm()):.2e} | |g| max {a.max():.2e} median {a.median():.2e} '
          f'frac<1e-8 {(a < 1e-8).float().mean():.3f} frac<1e-7 {(a < 1e-7).float().mean():.3f}', flush=True)
    if k == 0:
        # where do the gradient differences sit?  per channel group (local 0..127 / global 128..511) and relative per element
        for name, sl in (('local', slice(0, 128)), ('global', slice(128, 512))):
            print(f'   {name}: rel-L2 {float(dg[:, sl].norm() / gr[:, sl].norm()):.2e}  |g| median {a[:, sl].median():.2e}')
        big = a > a.median()
        print(f'   elements above the median |g|: max relative error {float((dg[big] / a[big]).max()):.2e}, mean {float((dg[big] / a[big]).mean()):.2e}')
d = (out - out_ref).abs()
print('out diff max', float(d.max()), 'mean', float(d.mean()), 'mean inside the hole', float(d[mask.repeat(1, 3, 1, 1) > 0].mean()))
