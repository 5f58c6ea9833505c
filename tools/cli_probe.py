#!/usr/bin/env python3
"""`python -m lama_amd.predict ... profile=true` on N synthetic 512 x 512 PNG pairs (tmpfs): where the main thread of the round loop spends its time.
usage: cli_probe.py [n_images=1536] [io_threads=8,16]"""
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
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1536
    threads = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else '8,16').split(',')]
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
        base = np.random.RandomState(7).randint(0, 256, (576, 576, 3)).astype('uint8')
        m = np.zeros((512, 512), 'uint8')
        m[128:384, 128:384] = 255
        mpng = encode_png(m)
        for i in range(n):
            oy, ox = (i * 7) % 64, (i * 13) % 64
            with open(os.path.join(indir, f'im{i:05d}.png'), 'wb') as f:
                f.write(encode_png(base[oy:oy + 512, ox:ox + 512]))
            with open(os.path.join(indir, f'im{i:05d}_mask001.png'), 'wb') as f:
                f.write(mpng)
        for T, X in [(t, x) for t in threads for x in ([[]] if not extra else [[], extra])]:
            od = os.path.join(root, f'out{T}')
            r = subprocess.run([sys.executable, '-m', 'lama_amd.predict', f'model.path={mdir}', f'indir={indir}', f'outdir={od}', f'io_threads={T}', 'profile=true'] + X,
                               cwd=ROOT, capture_output=True, text=True, timeout=900)
            print(f'== io_threads={T} {" ".join(X)}', flush=True)
            print('\n'.join(ln for ln in (r.stdout + r.stderr).splitlines() if ln.startswith(('wrote', 'main-thread')) or 'Error' in ln), flush=True)
            shutil.rmtree(od, ignore_errors=True)
    finally:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == '__main__':
    main()
