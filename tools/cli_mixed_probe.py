#!/usr/bin/env python3
"""`python -m lama_amd.predict ... profile=true` on a directory of DIFFERENTLY sized images (one or two per padded shape): what a new shape costs.
usage: cli_mixed_probe.py [n_shapes=16] [per_shape=1]"""
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from lama_amd import _lib as L  # noqa: E402
from lama_amd.predict import encode_png  # noqa: E402


def main():
    ns = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    per = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    extra = sys.argv[3:]
    model = bench.build_model(torch.device('cuda', 0), L.PREC_F16X3)
    root = tempfile.mkdtemp(prefix='lama_cli_', dir='/dev/shm')
    try:
        mdir, indir = os.path.join(root, 'model'), os.path.join(root, 'in')
        os.makedirs(os.path.join(mdir, 'models'))
        os.makedirs(indir)
        with open(os.path.join(mdir, 'config.yaml'), 'w') as f:
            yaml.safe_dump(dict(training_model=dict(kind='default', concat_mask=True), generator=dict(bench.BIG_LAMA)), f)
        torch.save({'state_dict': {k: v.detach().cpu() for k, v in model.state_dict().items()}}, os.path.join(mdir, 'models', 'best.ckpt'))
        del model
        torch.cuda.empty_cache()
        rng = np.random.RandomState(7)
        base = rng.randint(0, 256, (1600, 1600, 3)).astype('uint8')
        npx = 0
        for i in range(ns):
            h, w = 400 + 37 * i, 600 + 53 * i                      # 400 x 600 ... ~955 x 1395: every image its own padded shape
            for k in range(per):
                m = np.zeros((h, w), 'uint8')
                m[h // 4: h // 2, w // 4: w // 2] = 255
                with open(os.path.join(indir, f'im{i:03d}_{k}.png'), 'wb') as f:
                    f.write(encode_png(base[k:h + k, :w]))
                with open(os.path.join(indir, f'im{i:03d}_{k}_mask001.png'), 'wb') as f:
                    f.write(encode_png(m))
                npx += h * w
        od = os.path.join(root, 'out')
        r = subprocess.run([sys.executable, '-m', 'lama_amd.predict', f'model.path={mdir}', f'indir={indir}', f'outdir={od}', 'profile=true'] + extra,
                           cwd=ROOT, capture_output=True, text=True, timeout=900)
        print(f'== {ns} shapes x {per} image(s), {npx / 1e6:.1f} Mpixel', flush=True)
        print('\n'.join(ln for ln in (r.stdout + r.stderr).splitlines() if ln.startswith(('wrote', 'main-thread')) or 'Error' in ln), flush=True)
    finally:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == '__main__':
    main()
