"""The host SIMT emulator (tests/hipemu) checks itself where no kernel exercises it yet: the fp8 matrix instruction added as groundwork for the
fp16 + 2 x fp8 arithmetic.  Test infrastructure only."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'hipemu'))


def test_emulated_fp8_mfma_matches_scalar_reference(tmp_path):
    import build_emu
    cxx = build_emu._cxx()
    exe = str(tmp_path / 'selftest_fp8')
    root = os.path.dirname(HERE)
    cmd = [cxx, '-x', 'c++', '-std=c++17', '-O1', '-DLAMA_PROFILING', '-Wno-unknown-attributes', '-Wno-ignored-attributes',
           '-I' + os.path.join(HERE, 'hipemu'), '-I' + os.path.join(root, 'include'), '-I' + os.path.join(root, 'lama_amd', 'csrc'),
           os.path.join(HERE, 'hipemu', 'selftest_fp8.cpp'), os.path.join(HERE, 'hipemu', 'hipemu_runtime.cpp'), '-o', exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip().endswith('ok'), r.stdout + r.stderr
