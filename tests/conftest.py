import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """The CPU suite (-m "not gpu": 340 emulator / oracle / host tests, about half an hour on one core, 6.5 minutes on eight) runs on up to eight
    pytest-xdist workers (one per core) when xdist is installed and nothing else was asked for; the GPU suite never does (one process owns the GPU, and the co-residency tests
    must not share it)."""
    if os.environ.get('PYTEST_XDIST_WORKER') or hasattr(config, 'workerinput'):     # a worker runs this hook too: it must never spawn workers
        return None
    opt = config.option
    if getattr(opt, 'markexpr', '') != 'not gpu' or getattr(opt, 'numprocesses', 'absent') is not None or (os.cpu_count() or 1) < 4:
        return None
    if os.environ.get('LAMA_TEST_WORKERS', '') == '0' or not config.pluginmanager.hasplugin('xdist'):
        return None
    n = int(os.environ.get('LAMA_TEST_WORKERS', str(min(8, os.cpu_count() or 1))))
    opt.numprocesses, opt.dist, opt.tx = n, 'load', ['popen'] * n
    # one worker per core: torch / oneDNN / OpenMP inside a worker must not spawn a thread per core as well (8 x 8 threads on 8 cores made the
    # emulator tests 2-3x slower); the workers inherit the environment
    for var in ('OMP_NUM_THREADS', 'MKL_NUM_THREADS'):
        os.environ.setdefault(var, str(max(1, (os.cpu_count() or 1) // n)))
    return None


# the longest tests first: xdist's load balancing hands tests out in collection order, and a three-minute test that starts last is the wall time
_SLOW_FIRST = ('test_refine_oracle_pin', 'test_inplace_residual_and_aliased_t', 'test_bench_step_loop_world2', 'test_generator_fused_fft_path_and_layerwise',
               'test_infer_one_scale_matches_autograd_adam', 'test_biglama_shape_matches_reference', 'test_predict_world2_gloo', 'test_conv1_rides_in_the_global',
               'test_f16_split_overflow_falls_back', 'test_deferred_range_check_has_no_read_back', 'test_ffc_units_at_biglama_channel_counts',
               'test_deferred_winograd_output_transform_plan', 'test_refine_predict_two_scales', 'test_generator_fp16_activation_path',
               'test_predict_range_error_leaves', 'test_host_fed_step_double_buffering', 'test_predict_loop_with_scale_factor')


def pytest_collection_modifyitems(config, items):
    def rank(it):
        for i, name in enumerate(_SLOW_FIRST):
            if name in it.nodeid:
                return i
        return len(_SLOW_FIRST)
    items.sort(key=rank)          # stable: everything else keeps its order


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN
