"""bench.py's driver contract on the GPU box, through the path a multi-GPU run takes: `python bench.py --gpus N` without a
torch.distributed environment re-executes itself under torch.distributed.run, the ranks form an RCCL ('nccl') process group, every step
ends in the gather (to rank 0) of the u8 output images, the timed region is bracketed by barrier + synchronize and the time is the MAX over
ranks.  A single-GPU box can host one rank only (LAMA_BENCH_FORCE_DIST=1 takes that path with world_size 1); the world_size-2 logic is
covered on CPU by tests/test_dist_gloo.py."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_line_through_the_rccl_path():
    env = dict(os.environ, LAMA_BENCH_FORCE_DIST='1')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '3', '--warmup', '1', '--no-cpu-baseline',
                        '--no-f32-leg', '--no-eager-leg'], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]                    # ONE JSON line, from rank 0
    d = json.loads(lines[0])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype',
              'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in d, k
    assert d['n_gpus'] == 1 and d['n_ranks_seen'] == 1 and d['steps'] == 3 and d['warmup'] == 1
    assert d['unit'] == 'images/s' and d['higher_is_better'] is True and d['scaling'] == 'weak' and d['data'] == 'synthetic'
    assert 'workload' in d['config'] and 'model' not in d['config']
    assert abs(d['value'] - 8 * 3 / (d['ms_per_step'] * 3e-3)) < 0.01 * d['value']      # whole-job images / timed seconds
    rf = d['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert k in rf, k
    assert rf['bound'] in ('hbm', 'mfma') and abs(rf['frac'] - rf['achieved'] / rf['peak']) < 1e-3
