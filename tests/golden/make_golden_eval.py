"""Golden vectors of the SSIM evaluator, produced by the REFERENCE's own class (run in the build container only; /root/reference does
not exist on the GPU box).  usage: python tests/golden/make_golden_eval.py  ->  tests/golden/ssim.npz"""
import importlib.util
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location('ref_ssim', '/root/reference/saicinpainting/evaluation/losses/ssim.py')
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

g = torch.Generator().manual_seed(20260923)
out = {}
for tag, (b, c, h, w, ws) in dict(a=(2, 3, 40, 56, 11), b=(3, 1, 17, 33, 7), c=(1, 3, 64, 64, 11)).items():
    x = torch.rand(b, c, h, w, generator=g)
    y = (x + 0.15 * torch.randn(b, c, h, w, generator=g)).clamp(0, 1)
    if tag == 'c':
        y = x.clone()                      # identical images: SSIM = 1
    with torch.no_grad():
        per = ref.SSIM(window_size=ws, size_average=False)(x, y)
        mean = ref.SSIM(window_size=ws, size_average=True)(x, y)
    out[f'{tag}_x'], out[f'{tag}_y'] = x.numpy(), y.numpy()
    out[f'{tag}_ws'] = np.array(ws)
    out[f'{tag}_per_image'], out[f'{tag}_mean'] = per.numpy(), mean.numpy()
np.savez_compressed(os.path.join(HERE, 'ssim.npz'), **out)
print({k: v.shape for k, v in out.items()})
