"""CPU test of the drop-in boundary: the gfx950 shared object builds, loads without a GPU and exports every entry point that
include/lama_hip.h declares; the host binding (lama_amd/_lib.py) binds the same set.  No compute call is made here."""
import ctypes
import os
import re

from lama_amd import build as B

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, 'include', 'lama_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)          # comments cite reference lines, some contain parentheses
    src = re.sub(r'//[^\n]*', '', src)
    names = re.findall(r'^\s*(?:const\s+)?[A-Za-z_][A-Za-z0-9_]*\s*\*?\s+\*?\s*(lama_[a-z0-9_]+)\s*\(', src, flags=re.M)
    return sorted(set(names))


def test_header_declares_the_path_entry_points():
    names = _declared()
    for must in ('lama_version', 'lama_conv2d_pack_weight', 'lama_conv2d_fwd', 'lama_rfft2_fwd', 'lama_irfft2_fwd',
                 'lama_fourier_unit_fwd', 'lama_mask_compose_fwd', 'lama_blend_fwd', 'lama_quantize_u8_hwc_fwd'):
        assert must in names, (must, names)


def test_shared_object_exports_every_declared_symbol():
    lib = B.build(verbose=False)                               # no-op when the in-tree build is current
    dll = ctypes.CDLL(lib)                                     # loads on a GPU-less host: no HIP call at load time
    missing = [n for n in _declared() if not hasattr(dll, n)]
    assert not missing, missing
    dll.lama_version.restype = ctypes.c_int
    assert dll.lama_version() > 0                              # pure host function
    dll.lama_error_string.restype = ctypes.c_char_p
    dll.lama_error_string.argtypes = [ctypes.c_int]
    assert dll.lama_error_string(0)


def test_host_binding_covers_the_header():
    from lama_amd._lib import LamaLib
    lib = LamaLib(B.build(verbose=False))
    for n in _declared():
        assert getattr(lib._l, n) is not None
    # every entry point the binding CALLS has its argument types declared: an unset argtypes passes pointers as C ints (truncated to 32 bits --
    # round 4: a launch with truncated pointers is a memory fault on the GPU box and nothing on the CPU emulator tests that never reach it)
    src = open(os.path.join(ROOT, 'lama_amd', '_lib.py')).read()
    called = set(re.findall(r'self\._l\.(lama_[a-z0-9_]+)\(', src))
    untyped = sorted(n for n in called if getattr(lib._l, n).argtypes is None and n not in ('lama_version',))
    assert not untyped, untyped


def _gfx950_code_objects(path):
    """The gfx950 code objects bundled in a HIP shared object (clang offload bundles inside .hip_fatbin)."""
    import struct
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, 'fat.bin')
        subprocess.run(['objcopy', '-O', 'binary', '--only-section=.hip_fatbin', path, fat], check=True)
        data = open(fat, 'rb').read()
    magic = b'__CLANG_OFFLOAD_BUNDLE__'
    out, p = [], data.find(magic)
    while p >= 0:
        cnt = struct.unpack_from('<Q', data, p + 24)[0]
        off = p + 32
        for _ in range(cnt):
            o, sz, tl = struct.unpack_from('<QQQ', data, off)
            off += 24
            triple = data[off:off + tl].decode()
            off += tl
            if 'gfx950' in triple:
                out.append(data[p + o:p + o + sz])
        p = data.find(magic, p + 1)
    return out


def test_product_library_has_no_packed_fp32_instructions():
    """v_pk_{add,mul,fma}_f32 with an op_sel swizzle return wrong results on MI355X while another kernel's MFMA runs on
    the same SIMD (tools/race_probe9.py), so the shipped library must not contain packed-fp32 VALU instructions at all."""
    import subprocess
    import tempfile
    objdump = '/opt/rocm/lib/llvm/bin/llvm-objdump'
    if not os.path.exists(objdump):
        import pytest
        pytest.skip('llvm-objdump not available')
    cos = _gfx950_code_objects(B.build(verbose=False))
    assert len(cos) >= 6
    total_mfma = 0
    for co in cos:
        with tempfile.NamedTemporaryFile(suffix='.o') as f:
            f.write(co)
            f.flush()
            asm = subprocess.run([objdump, '-d', '--mcpu=gfx950', f.name], capture_output=True, text=True, check=True).stdout
        bad = re.findall(r'v_pk_(?:add|mul|fma)_f32[^\n]*', asm)
        assert not bad, bad[:3]
        total_mfma += asm.count('v_mfma')
    assert total_mfma > 1000          # the disassembly really covered the kernels
