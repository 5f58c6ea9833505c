"""Test helper: the kernel sources compiled for the host SIMT emulator (tests/hipemu)."""
import functools
import os
import sys

# small test shapes must reach the full-M 1x1 workgroups too (conv_wreg_host.inc reads this once)
os.environ.setdefault('LAMA_CW_1X1', '2')
os.environ.setdefault('LAMA_GEMM_WS', '2')
os.environ.setdefault('LAMA_STEM_WS', '2')
os.environ.setdefault('LAMA_HEAD_WS', '2')
# ... and the 12-wave all-rows workgroup of the global branch (production takes it from 160 tiles on: conv_wreg_host.inc)
os.environ.setdefault('LAMA_CW_G12', '2')

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'hipemu'))


@functools.lru_cache(None)
def emu_lib():
    import build_emu
    from lama_amd._lib import LamaLib
    return LamaLib(build_emu.build(), host_emulated=True)
