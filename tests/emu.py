"""Test helper: the kernel sources compiled for the host SIMT emulator (tests/hipemu)."""
import functools
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'hipemu'))


@functools.lru_cache(None)
def emu_lib():
    import build_emu
    from lama_amd._lib import LamaLib
    return LamaLib(build_emu.build())
