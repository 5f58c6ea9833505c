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
    return None


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN
