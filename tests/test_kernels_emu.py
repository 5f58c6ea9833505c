"""CPU tests of the HIP kernel SOURCES through the host SIMT emulator (tests/hipemu): index math, LDS
layouts, MFMA fragment maps, barrier placement and the C-ABI argument handling, checked against
plain torch fp32 ops.  The same comparisons run on the real GPU in test_kernels_gpu.py."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from lama_amd import _lib as L
from tests.emu import emu_lib


def _conv_ref(x, w, stride, pad, reflect, transposed, bias, act, resid, x2=None, w2=None, scale=None):
    if scale is not None:
        w = w * (scale[None, :, None, None] if transposed else scale[:, None, None, None])
    if transposed:
        y = F.conv_transpose2d(x, w, None, stride=2, padding=1, output_padding=1)
    else:
        xp = F.pad(x, (pad,) * 4, mode='reflect') if (pad and reflect) else F.pad(x, (pad,) * 4)
        y = F.conv2d(xp, w, None, stride=stride)
    if x2 is not None:
        y = y + F.conv2d(x2, w2)
    if bias is not None:
        y = y + bias[None, :, None, None]
    y = {0: lambda t: t, 1: torch.relu, 2: torch.sigmoid, 3: torch.tanh}[act](y)
    if resid is not None:
        y = y + resid
    return y


CONV_CASES = [
    # cin, cout, k, stride, pad, reflect, transposed, H, W, act, bias, resid, scale
    dict(cin=8, cout=16, k=3, stride=1, pad=1, H=12, W=20, act=1, bias=True, resid=True, scale=True),
    dict(cin=12, cout=40, k=3, stride=1, pad=1, H=9, W=7, act=0, bias=False, resid=False, scale=False),
    dict(cin=4, cout=8, k=7, stride=1, pad=3, H=16, W=24, act=1, bias=True, resid=False, scale=True),
    dict(cin=8, cout=3, k=7, stride=1, pad=3, H=16, W=16, act=2, bias=True, resid=False, scale=False),
    dict(cin=8, cout=16, k=3, stride=2, pad=1, H=16, W=24, act=1, bias=True, resid=False, scale=True),
    dict(cin=6, cout=16, k=3, stride=2, pad=1, H=10, W=14, act=1, bias=True, resid=False, scale=True),
    dict(cin=24, cout=12, k=1, stride=1, pad=0, H=6, W=11, act=1, bias=True, resid=False, scale=True),
    dict(cin=40, cout=72, k=1, stride=1, pad=0, H=16, W=9, act=0, bias=False, resid=True, scale=False),
    dict(cin=16, cout=8, k=3, stride=2, pad=1, H=8, W=12, act=1, bias=True, resid=False, scale=True, transposed=True),
    dict(cin=20, cout=36, k=3, stride=2, pad=1, H=5, W=7, act=0, bias=True, resid=False, scale=False, transposed=True),
    dict(cin=140, cout=130, k=3, stride=1, pad=1, H=8, W=8, act=1, bias=True, resid=True, scale=True),
    dict(cin=24, cout=192, k=1, stride=1, pad=0, H=9, W=16, act=1, bias=True, resid=False, scale=True),     # 192-row M tiles (bf16x3)
    dict(cin=20, cout=384, k=3, stride=1, pad=1, H=6, W=36, act=1, bias=True, resid=True, scale=False),
    dict(cin=24, cout=70, k=3, stride=2, pad=1, H=9, W=34, act=1, bias=True, resid=False, scale=True, transposed=True),   # fused classes, BM=128
    # weights-in-registers kernels (conv_wreg_dev.inc): 3x3 with cin % 32 == 0 / 1x1 with cin % 64 == 0, cout >= 96
    dict(cin=64, cout=130, k=3, stride=1, pad=1, H=7, W=37, act=1, bias=True, resid=True, scale=True),      # ragged tiles, 2 M tiles
    dict(cin=32, cout=96, k=3, stride=1, pad=1, H=5, W=6, act=0, bias=False, resid=False, scale=False),    # one chunk, narrow image
    dict(cin=128, cout=100, k=1, stride=1, pad=0, H=9, W=15, act=1, bias=True, resid=True, scale=True),    # flat 1x1, two chunks
    dict(cin=192, cout=200, k=1, stride=1, pad=0, H=12, W=11, act=2, bias=True, resid=False, scale=False),  # odd chunk count
    # ... the local conv's geometries by launch size (conv_wreg_host.inc), forced here with LAMA_CW_41: 3 = LAMA_CONV_COOPERATIVE's one
    # 4-wave workgroup per CU (32-channel chunks, six-deep A ring), 2 = two 4-wave workgroups per CU (16-channel chunks); row-rolling
    # and plain B reads each
    dict(cin=64, cout=128, k=3, stride=1, pad=1, H=7, W=37, act=1, bias=True, resid=True, scale=True, geo=3),
    dict(cin=96, cout=128, k=3, stride=1, pad=1, H=9, W=12, act=0, bias=False, resid=False, scale=False, geo=3),
    dict(cin=64, cout=256, k=3, stride=1, pad=1, H=7, W=37, act=1, bias=True, resid=True, scale=True, geo=2),
    dict(cin=96, cout=128, k=3, stride=1, pad=1, H=9, W=12, act=0, bias=False, resid=False, scale=False, geo=2),
    # ... ConvTranspose2d as ONE launch with all four parity classes in a wave (convt2_kernel, convt_dev.inc: cin % 32 == 0, cout % 64 == 0, W % 32 == 0, H % 4 == 0): one
    # tile / one chunk; 2 x 2 tiles, two chunks, two 64-row groups; and the fused LDS-staged launch on the same shape (LAMA_CT=0)
    dict(cin=32, cout=64, k=3, stride=2, pad=1, H=4, W=32, act=1, bias=True, resid=False, scale=True, transposed=True),
    dict(cin=64, cout=128, k=3, stride=2, pad=1, H=8, W=64, act=0, bias=False, resid=False, scale=False, transposed=True),
    dict(cin=64, cout=128, k=3, stride=2, pad=1, H=8, W=64, act=0, bias=False, resid=False, scale=False, transposed=True, ct=0),
    dict(cin=64, cout=128, k=3, stride=2, pad=1, H=8, W=64, act=2, bias=True, resid=False, scale=True, transposed=True),            # sigmoid: not convt2_kernel's epilogue -> the fused LDS-staged launch
    dict(cin=32, cout=64, k=3, stride=2, pad=1, H=5, W=37, act=1, bias=True, resid=False, scale=True, transposed=True),            # round 6: convt2_kernel with ragged last tiles (H % 4, W % 32 != 0)
    dict(cin=64, cout=64, k=3, stride=2, pad=1, H=6, W=70, act=0, bias=False, resid=False, scale=False, transposed=True, ct_grid=2),
    dict(cin=128, cout=64, k=3, stride=2, pad=1, H=12, W=32, act=1, bias=True, resid=False, scale=True, transposed=True, ct_grid=2),  # up3's channel counts: 8 sub-chunks, 6 tiles on 2 workgroups
    dict(cin=64, cout=128, k=3, stride=2, pad=1, H=8, W=64, act=1, bias=True, resid=False, scale=True, transposed=True, ct_grid=3),   # 16 tiles on 3 persistent workgroups
    # ... stride 2 (the downsampling convs): parity-split patch columns, five staging units per thread; odd sizes, ragged tiles, 2 M tiles
    dict(cin=32, cout=128, k=3, stride=2, pad=1, H=16, W=70, act=1, bias=True, resid=False, scale=True),
    dict(cin=64, cout=200, k=3, stride=2, pad=1, H=11, W=37, act=0, bias=False, resid=True, scale=False),
    # full-M 1x1 workgroups: 384 rows = 12 M-waves, 192 rows = 6 M-waves x 2 K-groups (K-group exchange + one residual set)
    dict(cin=128, cout=384, k=1, stride=1, pad=0, H=5, W=33, act=1, bias=True, resid=True, scale=True),
    dict(cin=192, cout=192, k=1, stride=1, pad=0, H=9, W=16, act=1, bias=True, resid=True, scale=True),
    dict(cin=64, cout=370, k=1, stride=1, pad=0, H=3, W=50, act=0, bias=False, resid=False, scale=False),   # ragged rows, one chunk
    # weights-stationary persistent GEMM (conv_ws_dev.inc): K = 192 / 384, rows in groups of 96, ragged last tile, 1 / 2 / 4 row groups
    dict(cin=384, cout=96, k=1, stride=1, pad=0, H=5, W=21, act=1, bias=True, resid=True, scale=True),
    dict(cin=192, cout=384, k=1, stride=1, pad=0, H=3, W=50, act=0, bias=False, resid=False, scale=False),
    dict(cin=384, cout=180, k=1, stride=1, pad=0, H=8, W=33, act=2, bias=True, resid=False, scale=True),      # rows past M in the last fragment
    # ... and over 64-pixel super-tiles of two interleaved MFMA tiles (gemm1x1_w4_kernel; LAMA_GEMM_W4=2 forces it at any launch size, LAMA_GEMM_W4_SLOTS
    # caps the workgroups per row group): super-tiles across the image boundary, groups past the batch, several rounds, left-over super-tiles, K = 192
    dict(cin=384, cout=96, k=1, stride=1, pad=0, H=7, W=38, act=1, bias=True, resid=True, scale=True, w4_slots=0),        # 532 pixels = 8.3 super-tiles, one round
    dict(cin=384, cout=96, k=1, stride=1, pad=0, H=7, W=38, act=1, bias=True, resid=True, scale=True, w4_slots=4),        # two rounds + one left-over super-tile
    dict(cin=384, cout=96, k=1, stride=1, pad=0, H=7, W=38, act=0, bias=False, resid=False, scale=False, w4_slots=2),     # four rounds + one left-over
    dict(cin=192, cout=192, k=1, stride=1, pad=0, H=6, W=40, act=1, bias=True, resid=False, scale=False, w4_slots=3),     # K = 192, two row groups, two rounds + 2 whole left-overs
    dict(cin=192, cout=96, k=1, stride=1, pad=0, H=6, W=40, act=1, bias=True, resid=True, scale=True, w4_slots=1),        # K = 192, eight rounds on one workgroup
    dict(cin=384, cout=384, k=1, stride=1, pad=0, H=4, W=33, act=1, bias=True, resid=True, scale=True, w4_slots=2),       # four row groups, 264 pixels
    # ... and with all of K in one wave (gemm1x1_wk_kernel, gemm_wk_dev.inc; LAMA_GEMM_WK=2 forces it at any launch size, LAMA_GEMM_WK_SLOTS caps the
    # workgroups): 384 -> 384 only
    dict(cin=384, cout=384, k=1, stride=1, pad=0, H=4, W=33, act=1, bias=True, resid=False, scale=True, wk_slots=0),      # 264 pixels = 4.1 super-tiles, one round, ragged last one
    dict(cin=384, cout=384, k=1, stride=1, pad=0, H=4, W=33, act=1, bias=True, resid=False, scale=True, wk_slots=4),      # one round + 1 left-over = 24 units on four fifth waves
    dict(cin=384, cout=384, k=1, stride=1, pad=0, H=4, W=33, act=0, bias=False, resid=False, scale=False, wk_slots=2),    # left-over too big for the fifth waves: 3 / 2 rounds
    dict(cin=384, cout=384, k=1, stride=1, pad=0, H=20, W=40, act=1, bias=True, resid=False, scale=False, wk_slots=12),   # 25 super-tiles: two rounds + 24 units, two per fifth wave
    dict(cin=384, cout=384, k=1, stride=1, pad=0, H=7, W=38, act=1, bias=True, resid=False, scale=True, wk_slots=8),      # HW = 266: an image boundary inside a super-tile
    # stem kernel (conv_stem_dev.inc): 7x7, cin <= 4, 33..64 output channels; ragged last segment, rows past M, image narrower than a segment
    dict(cin=4, cout=64, k=7, stride=1, pad=3, H=9, W=40, act=1, bias=True, resid=False, scale=True),
    dict(cin=3, cout=40, k=7, stride=1, pad=3, H=6, W=20, act=0, bias=False, resid=False, scale=False),
    # head kernel (conv_head_dev.inc): 7x7, 64 input channels, <= 4 output channels; two 26-column segments (ragged), rows not a multiple of 7
    dict(cin=64, cout=3, k=7, stride=1, pad=3, H=10, W=40, act=2, bias=True, resid=False, scale=False),
    dict(cin=64, cout=4, k=7, stride=1, pad=3, H=16, W=26, act=0, bias=False, resid=False, scale=True),
]


PRECISIONS = [L.PREC_F32, L.PREC_BF16X3, L.PREC_F16X3]
PREC_IDS = ['f32', 'bf16x3', 'f16x3']
# max-abs tolerance: the exact-fp32 MFMA path is an fmaf chain; the 3-term splits carry 16 (bf16) / 22 (f16) mantissa bits
CONV_TOL = {L.PREC_F32: dict(atol=2e-4, rtol=1e-4), L.PREC_BF16X3: dict(atol=6e-4, rtol=2e-4), L.PREC_F16X3: dict(atol=2e-4, rtol=1e-4)}


@pytest.mark.parametrize('prec', PRECISIONS, ids=PREC_IDS)
@pytest.mark.parametrize('case', CONV_CASES, ids=lambda c: f"k{c['k']}s{c['stride']}c{c['cin']}o{c['cout']}{'T' if c.get('transposed') else ''}{'geo%d' % c['geo'] if c.get('geo') else ''}{'w4s%d' % c['w4_slots'] if 'w4_slots' in c else ''}{'wks%d' % c['wk_slots'] if 'wk_slots' in c else ''}{'ct%d' % c['ct'] if 'ct' in c else ''}{'ctg%d' % c['ct_grid'] if 'ct_grid' in c else ''}")
def test_conv2d_emulated(case, prec, monkeypatch):
    lib = emu_lib()
    if 'ct' in case:
        monkeypatch.setenv('LAMA_CT', str(case['ct']))
    if 'ct_grid' in case:
        monkeypatch.setenv('LAMA_CT_GRID', str(case['ct_grid']))
    if 'w4_slots' in case:
        monkeypatch.setenv('LAMA_GEMM_W4', '2')
        monkeypatch.setenv('LAMA_GEMM_W4_SLOTS', str(case['w4_slots']))
    if 'wk_slots' in case:
        monkeypatch.setenv('LAMA_GEMM_WK', '2')
        monkeypatch.setenv('LAMA_GEMM_WK_SLOTS', str(case['wk_slots']))
    if case.get('geo'):
        monkeypatch.setenv('LAMA_CW_41', str(case['geo']))
    g = torch.Generator().manual_seed(1)
    B, cin, cout, k = 2, case['cin'], case['cout'], case['k']
    tr = case.get('transposed', False)
    x = torch.randn(B, cin, case['H'], case['W'], generator=g)
    w = torch.randn((cin, cout, k, k) if tr else (cout, cin, k, k), generator=g) * 0.2
    scale = torch.rand(cout, generator=g) + 0.5 if case['scale'] else None
    bias = torch.randn(cout, generator=g) if case['bias'] else None
    ref0 = _conv_ref(x, w, case['stride'], case['pad'], True, tr, None, 0, None, scale=scale)
    resid = torch.randn(ref0.shape, generator=g) if case['resid'] else None
    ref = _conv_ref(x, w, case['stride'], case['pad'], True, tr, bias, case['act'], resid, scale=scale)
    wp = lib.pack_conv_weight(w, scale, stride=case['stride'], transposed=tr, precision=prec)
    # write into a channel slice of a wider buffer to exercise the view arithmetic
    ybuf = torch.full((B, cout + 3, ref.shape[2], ref.shape[3]), 7.0)
    lib.conv2d(L.view(x), wp, L.view(ybuf, 2, cout), B, k, case['stride'], case['pad'],
               L.PAD_ZERO if tr else L.PAD_REFLECT, tr, bias, case['act'], None if resid is None else L.view(resid), precision=prec,
               cooperative=case.get('geo') == 3)
    y = ybuf[:, 2:2 + cout]
    assert torch.allclose(y, ref, **CONV_TOL[prec]), float((y - ref).abs().max())
    assert float(ybuf[:, :2].min()) == 7.0 and float(ybuf[:, -1].max()) == 7.0   # nothing written outside the view


@pytest.mark.parametrize('prec', PRECISIONS, ids=PREC_IDS)
def test_conv2d_fused_second_operand_emulated(prec):
    """out_g = relu(convl2g(x_l) + conv2(t) + b) + resid in one launch (ffc.py:161,223,253-254,288)."""
    lib = emu_lib()
    g = torch.Generator().manual_seed(2)
    B, cl, cg, half, H, W = 2, 8, 24, 12, 10, 12
    state = torch.randn(B, cl + cg, H, W, generator=g)
    t = torch.randn(B, half, H, W, generator=g)
    w1 = torch.randn(cg, cl, 3, 3, generator=g) * 0.2
    w2 = torch.randn(cg, half, 1, 1, generator=g) * 0.2
    scale, bias = torch.rand(cg, generator=g) + 0.5, torch.randn(cg, generator=g)
    ref = _conv_ref(state[:, :cl], w1, 1, 1, True, False, bias, 1, state[:, cl:], x2=t, w2=w2 * scale[:, None, None, None], scale=scale)
    out = torch.zeros_like(state)
    lib.conv2d(L.view(state, 0, cl), lib.pack_conv_weight(w1, scale, precision=prec), L.view(out, cl, cg), B, 3, 1, 1, L.PAD_REFLECT,
               False, bias, L.ACT_RELU, L.view(state, cl, cg), x2=L.view(t), w2_packed=lib.pack_conv_weight(w2, scale, precision=prec),
               precision=prec)
    assert torch.allclose(out[:, cl:], ref, **CONV_TOL[prec]), float((out[:, cl:] - ref).abs().max())


@pytest.mark.parametrize('cg_', [160, 384], ids=['cg160', 'cg384'])   # 384: all rows in one 12-wave workgroup
@pytest.mark.parametrize('prec', [L.PREC_BF16X3, L.PREC_F16X3], ids=['bf16x3', 'f16x3'])
def test_conv2d_fused_second_operand_wreg_emulated(prec, cg_):
    """Same fused launch at channel counts that take the weights-in-registers kernel (3x3 cin % 32 == 0 + 1x1 cin % 64 == 0)."""
    lib = emu_lib()
    g = torch.Generator().manual_seed(3)
    B, cl, cg, half, H, W = 2, 32, cg_, 64, 6, 35
    state = torch.randn(B, cl + cg, H, W, generator=g)
    t = torch.randn(B, half, H, W, generator=g)
    w1 = torch.randn(cg, cl, 3, 3, generator=g) * 0.2
    w2 = torch.randn(cg, half, 1, 1, generator=g) * 0.2
    scale, bias = torch.rand(cg, generator=g) + 0.5, torch.randn(cg, generator=g)
    ref = _conv_ref(state[:, :cl], w1, 1, 1, True, False, bias, 1, state[:, cl:], x2=t, w2=w2 * scale[:, None, None, None], scale=scale)
    out = torch.zeros_like(state)
    lib.conv2d(L.view(state, 0, cl), lib.pack_conv_weight(w1, scale, precision=prec), L.view(out, cl, cg), B, 3, 1, 1, L.PAD_REFLECT,
               False, bias, L.ACT_RELU, L.view(state, cl, cg), x2=L.view(t), w2_packed=lib.pack_conv_weight(w2, scale, precision=prec),
               precision=prec)
    assert torch.allclose(out[:, cl:], ref, **CONV_TOL[prec]), float((out[:, cl:] - ref).abs().max())
    assert float(out[:, :cl].abs().max()) == 0.0


def _spec_ref(x):
    ff = torch.fft.rfftn(x, dim=(-2, -1), norm='ortho')
    b, c, h, wf = ff.shape
    return torch.stack((ff.real, ff.imag), dim=2).reshape(b, 2 * c, h, wf)


def _inv_ref(spec, h, w):
    b, c2, _, wf = spec.shape
    s = spec.view(b, c2 // 2, 2, h, wf)
    return torch.fft.irfftn(torch.complex(s[:, :, 0], s[:, :, 1]), s=(h, w), dim=(-2, -1), norm='ortho')


FFT_SIZES = [(16, 16), (32, 32), (64, 64), (32, 64), (64, 16), (128, 32), (128, 128),   # fused LDS path (64 / 128 squares: one-buffer kernels)
             (8, 12), (5, 9), (10, 7), (24, 40), (17, 16), (13, 13),          # round 6: mixed-radix plane-in-LDS kernels (fft_mr_dev.inc): radix 2 / 3 / 4 / 5 / 7 / 8
             (45, 60), (21, 94), (27, 25),                                    # ... 9x5 / 4x3x5, 3x7 / 2x47 (thread-per-output pass of a prime), 3x3x3 / 5x5
             (9, 14), (35, 30), (22, 26), (1, 16), (3, 2), (49, 56), (67, 40),  # ... 7x2, 5x7 / 2x3x5, primes 11 / 13, degenerate planes, 7x7 / 8x7, a prime height
             (90, 160),                                                       # ... the bottleneck plane of a 720 x 1280 frame (512 threads)
             (53, 24), (20, 59), (61, 67), (106, 12), (9, 118),               # ... large prime factors (the 2 x 2-blocked pair-symmetric first pass, mr_pass_prime1): prime heights / widths, both, 2 x 53, 2 x 59
             (256, 32), (16, 512), (256, 256)]                                # two-pass LDS path


@pytest.mark.parametrize('hw', FFT_SIZES, ids=lambda s: f'{s[0]}x{s[1]}')
def test_rfft2_irfft2_emulated(hw):
    lib = emu_lib()
    h, w = hw
    g = torch.Generator().manual_seed(h * 131 + w)
    B, Cn = 2, 3
    wide = torch.randn(B, Cn + 2, h, w, generator=g)          # view = channels 1..Cn of a wider buffer
    x = wide[:, 1:1 + Cn]
    spec = torch.zeros(B, 2 * Cn, h, w // 2 + 1)
    nws = lib.fft_workspace_bytes(B, Cn, h, w)
    ws = torch.zeros(max(nws, 4) // 4)
    lib.rfft2(L.view(wide, 1, Cn), L.view(spec), B, ws)
    ref = _spec_ref(x)
    tol = 3e-5 if max(h, w) <= 128 else 1e-4
    assert torch.allclose(spec, ref, atol=tol, rtol=1e-4), float((spec - ref).abs().max())
    # inverse on a NON-Hermitian spectrum (as after conv+BN+ReLU), fused residual add
    spec2 = torch.relu(torch.randn(B, 2 * Cn, h, w // 2 + 1, generator=g))
    resid = torch.randn(B, Cn, h, w, generator=g)
    y = torch.zeros(B, Cn, h, w)
    lib.irfft2(L.view(spec2), L.view(resid), L.view(y), B, ws)
    ref2 = resid + _inv_ref(spec2, h, w)
    assert torch.allclose(y, ref2, atol=tol, rtol=1e-4), float((y - ref2).abs().max())
    y2 = torch.zeros(B, Cn, h, w)
    lib.irfft2(L.view(spec2), None, L.view(y2), B, ws)
    assert torch.allclose(y2, ref2 - resid, atol=tol, rtol=1e-4)


@pytest.mark.parametrize('hw', [(45, 60), (21, 94), (27, 25), (9, 14), (67, 40), (16, 24), (53, 24), (20, 59), (61, 67)], ids=lambda s: f'{s[0]}x{s[1]}')
def test_rfft2_irfft2_two_launch_mixed_radix_emulated(hw, monkeypatch):
    """Round 6: the two-launch form of the mixed-radix passes (mr2_* kernels: planes that do not fit one workgroup's LDS and are not powers of
    two), forced here on small planes (LAMA_FFT_MR=2, profiling build)."""
    monkeypatch.setenv('LAMA_FFT_MR', '2')
    lib = emu_lib()
    h, w = hw
    g = torch.Generator().manual_seed(h * 31 + w)
    B, Cn = 2, 3
    wide = torch.randn(B, Cn + 2, h, w, generator=g)
    x = wide[:, 1:1 + Cn]
    spec = torch.zeros(B, 2 * Cn, h, w // 2 + 1)
    ws = torch.zeros(max(lib.fft_workspace_bytes(B, Cn, h, w), 4) // 4)
    lib.rfft2(L.view(wide, 1, Cn), L.view(spec), B, ws)
    assert torch.allclose(spec, _spec_ref(x), atol=3e-5, rtol=1e-4)
    spec2 = torch.relu(torch.randn(B, 2 * Cn, h, w // 2 + 1, generator=g))
    resid = torch.randn(B, Cn, h, w, generator=g)
    y = torch.zeros(B, Cn, h, w)
    lib.irfft2(L.view(spec2), L.view(resid), L.view(y), B, ws)
    assert torch.allclose(y, resid + _inv_ref(spec2, h, w), atol=3e-5, rtol=1e-4)
    with pytest.raises(L.LamaError):
        lib.rfft2(L.view(wide, 1, Cn), L.view(spec), B, None)          # this form needs the workspace


def test_fft_masked_entries_emulated():
    """lama_rfft2_masked_fwd / lama_irfft2_masked_fwd (v108): the transform times [mask > 0] in one launch on 256 x 256 planes, LAMA_ERR_UNSUPPORTED
    (nothing launched) elsewhere."""
    lib = emu_lib()
    g = torch.Generator().manual_seed(77)
    B, Cn, h, w = 1, 2, 256, 256
    x = torch.randn(B, Cn, h, w, generator=g)
    ws = torch.zeros(max(lib.fft_workspace_bytes(B, Cn, h, w), 4) // 4)
    ms = torch.randn(B, 2 * Cn, h, w // 2 + 1, generator=g)
    spec, spec_m = torch.zeros(B, 2 * Cn, h, w // 2 + 1), torch.full((B, 2 * Cn, h, w // 2 + 1), 3.0)
    lib.rfft2(L.view(x), L.view(spec), B, ws)
    lib.rfft2(L.view(x), L.view(spec_m), B, ws, mask=L.view(ms))
    assert torch.equal(spec_m, spec * (ms > 0))
    spec2 = torch.relu(torch.randn(B, 2 * Cn, h, w // 2 + 1, generator=g))
    resid, my = torch.randn(B, Cn, h, w, generator=g), torch.randn(B, Cn, h, w, generator=g)
    y, y_m = torch.zeros(B, Cn, h, w), torch.full((B, Cn, h, w), 3.0)
    lib.irfft2(L.view(spec2), L.view(resid), L.view(y), B, ws)
    lib.irfft2(L.view(spec2), L.view(resid), L.view(y_m), B, ws, mask=L.view(my))
    assert torch.equal(y_m, y * (my > 0))
    xs, ss = torch.randn(1, 2, 64, 64, generator=g), torch.zeros(1, 4, 64, 33)
    with pytest.raises(L.LamaError) as ei:
        lib.rfft2(L.view(xs), L.view(ss), 1, ws, mask=L.view(torch.ones(1, 4, 64, 33)))
    assert ei.value.code == L.ERR_UNSUPPORTED and float(ss.abs().max()) == 0.0


@pytest.mark.parametrize('hw_seq', [(64, 1), (64, 2), (64, 3), (128, 3)], ids=lambda s: f'{s[0]}x{s[0]}seq{s[1]}')
def test_fft_sequential_planes_emulated(hw_seq, monkeypatch):
    """Sized one-plane kernels walking SEQ consecutive planes per workgroup with the next plane prefetched (LAMA_FFT_SEQ)."""
    lib = emu_lib()
    n, seq = hw_seq
    monkeypatch.setenv('LAMA_FFT_SEQ', str(seq))
    monkeypatch.setenv('LAMA_FFT_INPLACE', '0')     # the one-buffer 64 x 64 kernels (default) are covered by test_rfft2_irfft2*
    g = torch.Generator().manual_seed(n + seq)
    B, Cn = 2, 3
    wide = torch.randn(B, Cn + 1, n, n, generator=g)
    x = wide[:, 1:]
    spec = torch.zeros(B, 2 * Cn, n, n // 2 + 1)
    lib.rfft2(L.view(wide, 1, Cn), L.view(spec), B, None)
    ref = _spec_ref(x)
    assert torch.allclose(spec, ref, atol=3e-5, rtol=1e-4), float((spec - ref).abs().max())
    spec2 = torch.relu(torch.randn(B, 2 * Cn, n, n // 2 + 1, generator=g))
    resid = torch.randn(B, Cn, n, n, generator=g)
    for r in (resid, None):
        y = torch.zeros(B, Cn, n, n)
        lib.irfft2(L.view(spec2), None if r is None else L.view(r), L.view(y), B, None)
        ref2 = _inv_ref(spec2, n, n) + (0 if r is None else r)
        assert torch.allclose(y, ref2, atol=3e-5, rtol=1e-4), float((y - ref2).abs().max())


def test_fourier_unit_emulated():
    from oracle import lama_oracle as O
    lib = emu_lib()
    g = torch.Generator().manual_seed(5)
    for (B, Cn, h, w) in [(2, 6, 16, 16), (1, 4, 10, 12)]:
        x = torch.randn(B, Cn, h, w, generator=g)
        sd = {'fu.conv_layer.weight': torch.randn(2 * Cn, 2 * Cn, 1, 1, generator=g) * 0.3,
              'fu.bn.weight': torch.rand(2 * Cn, generator=g) + 0.5, 'fu.bn.bias': torch.randn(2 * Cn, generator=g) * 0.2,
              'fu.bn.running_mean': torch.randn(2 * Cn, generator=g) * 0.1, 'fu.bn.running_var': torch.rand(2 * Cn, generator=g) + 0.5}
        ref = O.fourier_unit(x, sd, 'fu')
        scale = sd['fu.bn.weight'] / torch.sqrt(sd['fu.bn.running_var'] + 1e-5)
        shift = sd['fu.bn.bias'] - sd['fu.bn.running_mean'] * scale
        wp = lib.pack_conv_weight(sd['fu.conv_layer.weight'], scale)
        ws = torch.zeros(lib.fourier_unit_workspace_bytes(B, Cn, h, w) // 4 + 1)
        y = torch.zeros_like(x)
        lib.fourier_unit(L.view(x), wp, shift, L.view(y), B, True, ws)
        assert torch.allclose(y, x + ref, atol=1e-4, rtol=1e-4), float((y - x - ref).abs().max())


def test_elementwise_emulated():
    lib = emu_lib()
    g = torch.Generator().manual_seed(9)
    B, H, W = 2, 10, 14
    img, mask = torch.rand(B, 3, H, W, generator=g), (torch.rand(B, 1, H, W, generator=g) > 0.5).float()
    pred = torch.rand(B, 3, H, W, generator=g)
    out = torch.zeros(B, 4, H, W)
    lib.mask_compose(L.view(img), L.view(mask), L.view(out), B)
    assert torch.equal(out, torch.cat([img * (1 - mask), mask], 1))
    bl = torch.zeros(B, 3, H, W)
    lib.blend(L.view(img), L.view(mask), L.view(pred), L.view(bl), B)
    assert torch.allclose(bl, mask * pred + (1 - mask) * img, atol=1e-7)
    u8 = torch.zeros(B, 7, 9, 3, dtype=torch.uint8)
    src = torch.rand(B, 3, H, W, generator=g) * 1.2 - 0.1
    lib.quantize_u8_hwc(L.view(src), u8, B, 7, 9)
    ref = np.clip(src.permute(0, 2, 3, 1).numpy()[:, :7, :9] * 255, 0, 255).astype('uint8')
    assert np.array_equal(u8.numpy(), ref)


def test_c_abi_rejects_bad_arguments():
    lib = emu_lib()
    x = torch.zeros(1, 4, 8, 8)
    with pytest.raises(L.LamaError):
        lib.pack_conv_weight(torch.zeros(4, 4, 5, 5), None)            # unsupported kernel size
    wp = lib.pack_conv_weight(torch.zeros(8, 4, 3, 3), None)
    with pytest.raises(L.LamaError):
        lib.conv2d(L.view(x), wp, L.view(torch.zeros(1, 8, 7, 8)), 1, 3, 1, 1)    # wrong output shape
    with pytest.raises(L.LamaError):
        lib.rfft2(L.view(x), L.view(torch.zeros(1, 8, 8, 4)), 1)      # wrong spectrum width
    with pytest.raises(L.LamaError):
        lib.rfft2(L.view(torch.zeros(1, 4, 192, 256)), L.view(torch.zeros(1, 8, 192, 129)), 1, None)   # missing workspace (a plane too large for one workgroup's LDS: two launches)


def test_f16_split_range_watch_emulated():
    """lama_conv2d_args.range_flag: an activation beyond 65504 (or a NaN) met while the f16 split is applied raises the flag;
    in-range inputs and the other precisions leave it alone (conv_split3.inc cb_split2 / cb_range_report)."""
    lib = emu_lib()
    g = torch.Generator().manual_seed(31)
    for (cin, cout, k, H, W) in [(8, 16, 3, 8, 8), (64, 96, 3, 5, 6), (192, 96, 1, 4, 16), (4, 64, 7, 9, 40), (64, 3, 7, 10, 40)]:
        x = torch.randn(1, cin, H, W, generator=g)
        w = torch.randn(cout, cin, k, k, generator=g) * 0.1
        y = torch.zeros(1, cout, H, W)
        for prec in (L.PREC_F16X3, L.PREC_BF16X3):
            wp = lib.pack_conv_weight(w, None, precision=prec)
            for big, want in ((None, 0), (7.0e4, 1), (-1.0e5, 1), (float('nan'), 1), (6.5e4, 0)):
                xx = x.clone()
                if big is not None:
                    xx[0, cin // 2, H // 2, W // 3] = big
                flag = torch.zeros(1, dtype=torch.int32)
                lib.conv2d(L.view(xx), wp, L.view(y), 1, k, 1, k // 2, L.PAD_REFLECT, False, None, L.ACT_NONE, precision=prec, range_flag=flag)
                assert int(flag) == (want if prec == L.PREC_F16X3 else 0), (cin, cout, k, prec, big, int(flag))
    with pytest.raises(L.LamaRangeError):                                  # weights are checked by the host at pack time
        lib.pack_conv_weight(torch.full((8, 8, 1, 1), 7.0e4), None, precision=L.PREC_F16X3)
    lib.pack_conv_weight(torch.full((8, 8, 1, 1), 7.0e4), None, precision=L.PREC_BF16X3)


@pytest.mark.parametrize('hw', [(64, 64), (128, 128), (32, 32), (16, 32), (10, 12), (256, 32)], ids=lambda s: f'{s[0]}x{s[1]}')
def test_rfft2_irfft2_fp16_io_emulated(hw):
    """LAMA_DT_F16 tensors (BASELINE configs[2]): fp16 planes / spectra in memory, fp32 arithmetic inside the kernel.  Against the
    fp32 reference on the fp16-rounded inputs, within fp16 output rounding."""
    lib = emu_lib()
    h, w = hw
    g = torch.Generator().manual_seed(h * 7 + w)
    B, Cn = 2, 4
    wide = torch.randn(B, Cn + 1, h, w, generator=g).half()
    x = wide[:, 1:].float()
    spec = torch.zeros(B, 2 * Cn, h, w // 2 + 1, dtype=torch.float16)
    ws = torch.zeros(max(lib.fft_workspace_bytes(B, Cn, h, w), 4) // 4)
    lib.rfft2(L.view(wide, 1, Cn), L.view(spec), B, ws)
    ref = _spec_ref(x)
    assert torch.allclose(spec.float(), ref, atol=2e-3 * float(ref.abs().max()), rtol=2e-3), float((spec.float() - ref).abs().max())
    spec2 = torch.relu(torch.randn(B, 2 * Cn, h, w // 2 + 1, generator=g)).half()
    resid = torch.randn(B, Cn, h, w, generator=g).half()
    y = torch.zeros(B, Cn, h, w, dtype=torch.float16)
    lib.irfft2(L.view(spec2), L.view(resid), L.view(y), B, ws)
    ref2 = resid.float() + _inv_ref(spec2.float(), h, w)
    assert torch.allclose(y.float(), ref2, atol=4e-3, rtol=2e-3), float((y.float() - ref2).abs().max())
    with pytest.raises(L.LamaError):
        lib.rfft2(L.view(wide, 1, Cn), L.view(spec.float()), B, ws)          # mixed element types


def _conv_f16_ref(x, w, stride, pad, reflect, transposed, bias, act, resid, x2=None, w2=None, scale=None):
    """LAMA_PREC_F16: fp16 activations, BatchNorm-folded weights as hi + lo fp16 parts (two products per MAC: 22 mantissa bits of the
    weights survive), exact products, fp32 accumulation / epilogue."""
    if scale is not None:
        w = w * (scale[None, :, None, None] if transposed else scale[:, None, None, None])
        if w2 is not None:
            w2 = w2 * scale[:, None, None, None]
    return _conv_ref(x.float(), w, stride, pad, reflect, transposed, bias, act, None if resid is None else resid.float(),
                     x2=None if x2 is None else x2.float(), w2=w2)


F16_CASES = [
    # persistent pointwise GEMM (conv_ws_dev.inc)
    dict(cin=384, cout=96, k=1, stride=1, pad=0, H=5, W=21, act=1, bias=True, resid=True, scale=True),
    dict(cin=192, cout=384, k=1, stride=1, pad=0, H=3, W=50, act=0, bias=False, resid=False, scale=False),
    dict(cin=384, cout=180, k=1, stride=1, pad=0, H=8, W=33, act=2, bias=True, resid=False, scale=True),
    # weights-in-registers kernels (conv_wreg_dev.inc)
    dict(cin=64, cout=130, k=3, stride=1, pad=1, H=7, W=37, act=1, bias=True, resid=True, scale=True),
    dict(cin=32, cout=96, k=3, stride=1, pad=1, H=5, W=6, act=0, bias=False, resid=False, scale=False),
    dict(cin=128, cout=100, k=1, stride=1, pad=0, H=9, W=15, act=1, bias=True, resid=True, scale=True),
    dict(cin=128, cout=384, k=1, stride=1, pad=0, H=5, W=33, act=1, bias=True, resid=True, scale=True),
    dict(cin=192, cout=192, k=1, stride=1, pad=0, H=9, W=16, act=1, bias=True, resid=True, scale=True),
    dict(cin=32, cout=128, k=3, stride=2, pad=1, H=16, W=70, act=1, bias=True, resid=False, scale=True),       # ... stride 2
    # LDS-staged kernel (conv_split3.inc): 3x3, stride 2, transposed, 1x1 fallbacks, channel tails
    dict(cin=8, cout=16, k=3, stride=1, pad=1, H=12, W=20, act=1, bias=True, resid=True, scale=True),
    dict(cin=12, cout=40, k=3, stride=1, pad=1, H=9, W=7, act=0, bias=False, resid=False, scale=False),
    dict(cin=8, cout=16, k=3, stride=2, pad=1, H=16, W=24, act=1, bias=True, resid=False, scale=True),
    dict(cin=6, cout=16, k=3, stride=2, pad=1, H=10, W=14, act=1, bias=True, resid=False, scale=True),
    dict(cin=24, cout=12, k=1, stride=1, pad=0, H=6, W=11, act=1, bias=True, resid=False, scale=True),
    dict(cin=16, cout=8, k=3, stride=2, pad=1, H=8, W=12, act=1, bias=True, resid=False, scale=True, transposed=True),
    dict(cin=20, cout=36, k=3, stride=2, pad=1, H=5, W=7, act=0, bias=True, resid=False, scale=False, transposed=True),
    dict(cin=24, cout=70, k=3, stride=2, pad=1, H=9, W=34, act=1, bias=True, resid=False, scale=True, transposed=True),
    dict(cin=140, cout=130, k=3, stride=1, pad=1, H=8, W=8, act=1, bias=True, resid=True, scale=True),
]
# the two ends of the fp16 path: the stem reads the fp32 network input, the head writes the fp32 image
F16_STEM_HEAD = [
    (dict(cin=4, cout=64, k=7, stride=1, pad=3, H=9, W=40, act=1, bias=True, resid=False, scale=True), torch.float32, torch.float16),
    (dict(cin=64, cout=3, k=7, stride=1, pad=3, H=10, W=40, act=2, bias=True, resid=False, scale=False), torch.float16, torch.float32),
]


def _run_f16_case(lib, case, x_dtype=torch.float16, y_dtype=torch.float16):
    g = torch.Generator().manual_seed(3)
    B, cin, cout, k = 2, case['cin'], case['cout'], case['k']
    tr = case.get('transposed', False)
    x = torch.randn(B, cin, case['H'], case['W'], generator=g).to(x_dtype)
    w = torch.randn((cin, cout, k, k) if tr else (cout, cin, k, k), generator=g) * 0.2
    scale = torch.rand(cout, generator=g) + 0.5 if case['scale'] else None
    bias = torch.randn(cout, generator=g) if case['bias'] else None
    ref0 = _conv_f16_ref(x.half(), w, case['stride'], case['pad'], True, tr, None, 0, None, scale=scale)
    resid = torch.randn(ref0.shape, generator=g).to(y_dtype) if case['resid'] else None
    ref = _conv_f16_ref(x.half(), w, case['stride'], case['pad'], True, tr, bias, case['act'], resid, scale=scale)   # fp32 inputs are rounded once
    wp = lib.pack_conv_weight(w, scale, stride=case['stride'], transposed=tr, precision=L.PREC_F16)
    ybuf = torch.full((B, cout + 3, ref.shape[2], ref.shape[3]), 7.0, dtype=y_dtype)
    lib.conv2d(L.view(x), wp, L.view(ybuf, 2, cout), B, k, case['stride'], case['pad'], L.PAD_ZERO if tr else L.PAD_REFLECT, tr, bias,
               case['act'], None if resid is None else L.view(resid), precision=L.PREC_F16)
    y = ybuf[:, 2:2 + cout].float()
    tol = 2e-3 * max(1.0, float(ref.abs().max())) if y_dtype == torch.float16 else 1e-4     # one fp16 rounding of the output
    assert float((y - ref).abs().max()) < tol, float((y - ref).abs().max())
    assert float(ybuf[:, :2].float().min()) == 7.0 and float(ybuf[:, -1].float().max()) == 7.0


@pytest.mark.parametrize('case', F16_CASES, ids=lambda c: f"k{c['k']}s{c['stride']}c{c['cin']}o{c['cout']}{'T' if c.get('transposed') else ''}")
def test_conv2d_fp16_io_emulated(case):
    _run_f16_case(emu_lib(), case)


# LAMA_PREC_F16 keeps the resnet blocks' residual stream in fp32: the launches that read or write it mix
# element types -- (x dtype, y/resid dtype); every kernel family has to take both directions or fall through to one that does
F16_MIXED = [
    dict(cin=64, cout=130, k=3, stride=1, pad=1, H=7, W=37, act=1, bias=True, resid=True, scale=True),         # weights-in-registers, local
    dict(cin=384, cout=96, k=1, stride=1, pad=0, H=5, W=21, act=1, bias=True, resid=False, scale=True),         # persistent GEMM (conv1)
    dict(cin=384, cout=180, k=1, stride=1, pad=0, H=8, W=33, act=1, bias=True, resid=True, scale=True),         # ... with a residual: falls through
    dict(cin=128, cout=100, k=1, stride=1, pad=0, H=9, W=15, act=1, bias=True, resid=True, scale=True),         # 1x1 wreg shape: falls through
    dict(cin=64, cout=200, k=3, stride=2, pad=1, H=11, W=37, act=1, bias=True, resid=False, scale=True),        # weights-in-registers, stride 2 (last downsample)
    dict(cin=8, cout=16, k=3, stride=2, pad=1, H=16, W=24, act=1, bias=True, resid=False, scale=True),          # LDS-staged, the last downsample
    dict(cin=12, cout=40, k=3, stride=1, pad=1, H=9, W=7, act=0, bias=False, resid=True, scale=False),
    dict(cin=24, cout=70, k=3, stride=2, pad=1, H=9, W=34, act=1, bias=True, resid=False, scale=True, transposed=True),   # first upsample
]


@pytest.mark.parametrize('xdt,ydt', [(torch.float32, torch.float16), (torch.float16, torch.float32)], ids=['f32_to_f16', 'f16_to_f32'])
@pytest.mark.parametrize('case', F16_MIXED, ids=lambda c: f"k{c['k']}s{c['stride']}c{c['cin']}o{c['cout']}{'T' if c.get('transposed') else ''}")
def test_conv2d_fp16_mixed_io_emulated(case, xdt, ydt):
    _run_f16_case(emu_lib(), case, xdt, ydt)


@pytest.mark.parametrize('sdt,odt', [(torch.float16, torch.float16), (torch.float32, torch.float16), (torch.float16, torch.float32)],
                         ids=['f16', 'state_f32', 'out_f32'])
@pytest.mark.parametrize('cg_', [160, 384], ids=['cg160', 'cg384'])
def test_conv2d_fp16_fused_second_operand_wreg_emulated(cg_, sdt, odt):
    """The bottleneck global-branch launch (3x3 over x_l + 1x1 over t + bias + ReLU + residual) with fp16 activations; the block's
    first layer reads the fp32 residual stream (state fp32, t and the output fp16), its second layer writes it (x and t fp16, the
    residual and the output fp32)."""
    lib = emu_lib()
    g = torch.Generator().manual_seed(3)
    B, cl, cg, half, H, W = 2, 32, cg_, 64, 6, 35
    state = torch.randn(B, cl + cg, H, W, generator=g).to(sdt)
    t = torch.randn(B, half, H, W, generator=g).half()
    w1 = torch.randn(cg, cl, 3, 3, generator=g) * 0.2
    w2 = torch.randn(cg, half, 1, 1, generator=g) * 0.2
    scale, bias = torch.rand(cg, generator=g) + 0.5, torch.randn(cg, generator=g)
    resid = torch.randn(B, cg, H, W, generator=g).to(odt)
    ref = _conv_f16_ref(state[:, :cl].half(), w1, 1, 1, True, False, bias, 1, resid, x2=t, w2=w2, scale=scale)
    out = torch.zeros(B, cl + cg, H, W, dtype=odt)
    lib.conv2d(L.view(state, 0, cl), lib.pack_conv_weight(w1, scale, precision=L.PREC_F16), L.view(out, cl, cg), B, 3, 1, 1, L.PAD_REFLECT,
               False, bias, L.ACT_RELU, L.view(resid), x2=L.view(t), w2_packed=lib.pack_conv_weight(w2, scale, precision=L.PREC_F16),
               precision=L.PREC_F16)
    err = float((out[:, cl:].float() - ref).abs().max())
    assert err < (2e-3 if odt == torch.float16 else 1e-4) * max(1.0, float(ref.abs().max())), err
    assert float(out[:, :cl].float().abs().max()) == 0.0


@pytest.mark.parametrize('case,xdt,ydt', F16_STEM_HEAD, ids=['stem_f32_to_f16', 'head_f16_to_f32'])
def test_conv2d_fp16_stem_head_emulated(case, xdt, ydt):
    _run_f16_case(emu_lib(), case, xdt, ydt)


# Winograd F(2x2, 3x3) form of the stride-1 3x3 reflect conv (wino_dev.inc): W = 32 / 64 (bands of 8 / 4 tile rows), one and two bands,
# one and two 128-row groups, 32 and 64 input channels (one / two chunks), with and without bias / activation / residual
WINO_CASES = [
    dict(cin=32, cout=128, H=16, W=32, B=1, act=1, bias=True, resid=True, scale=True),
    dict(cin=64, cout=128, H=16, W=64, B=2, act=0, bias=False, resid=False, scale=False),
    dict(cin=32, cout=256, H=8, W=64, B=1, act=1, bias=True, resid=False, scale=True),
    dict(cin=128, cout=128, H=16, W=32, B=1, act=1, bias=True, resid=True, scale=True),       # 8 (band, xi) units: the input channels dealt to four workgroups each (split-K of small launches)
    # round 6: any plane size -- tile rows cut into overlapping segments of 8 / 16 / 32 column quads, reflected columns inside the last quad, odd heights
    dict(cin=32, cout=128, H=9, W=30, B=1, act=1, bias=True, resid=True, scale=True),         # one segment of 8 quads, W % 4 = 2, odd H
    dict(cin=32, cout=128, H=10, W=37, B=2, act=0, bias=True, resid=False, scale=False),      # two segments, W % 4 = 1 (the reflected column is the left neighbour's)
    dict(cin=32, cout=128, H=7, W=43, B=1, act=1, bias=False, resid=True, scale=True),        # W % 4 = 3
    dict(cin=32, cout=128, H=6, W=68, B=1, act=1, bias=True, resid=True, scale=False),        # three segments of 8 quads for 17
    dict(cin=32, cout=128, H=4, W=136, B=1, act=1, bias=True, resid=False, scale=True, qr=16),  # forced 16-quad segments: three for 34
    dict(cin=32, cout=128, H=4, W=136, B=1, act=0, bias=False, resid=True, scale=False, qr=32), # forced 32-quad segments: two for 34
]


@pytest.mark.parametrize('prec', [L.PREC_BF16X3, L.PREC_F16X3], ids=['bf16x3', 'f16x3'])
@pytest.mark.parametrize('case', WINO_CASES, ids=lambda c: f"c{c['cin']}o{c['cout']}_{c['H']}x{c['W']}b{c['B']}")
def test_winograd_conv3x3_emulated(case, prec, monkeypatch):
    """lama_winograd_conv3x3_fwd against the plain torch conv (reflect pad 1) -- the same function as lama_conv2d_fwd on this layer --
    and against the direct HIP kernel on the same inputs."""
    lib = emu_lib()
    if case.get('qr'):
        monkeypatch.setenv('LAMA_WG_QR', str(case['qr']))
    g = torch.Generator().manual_seed(5)
    B, cin, cout, H, W = case['B'], case['cin'], case['cout'], case['H'], case['W']
    x = torch.randn(B, cin + 2, H, W, generator=g)[:, 1:1 + cin]            # a channel slice of a wider buffer
    xbuf = torch.zeros(B, cin + 2, H, W)
    xbuf[:, 1:1 + cin] = x
    w = torch.randn(cout, cin, 3, 3, generator=g) * 0.2
    scale = torch.rand(cout, generator=g) + 0.5 if case['scale'] else None
    bias = torch.randn(cout, generator=g) if case['bias'] else None
    ref0 = _conv_ref(x, w, 1, 1, True, False, None, 0, None, scale=scale)
    resid = torch.randn(ref0.shape, generator=g) if case['resid'] else None
    ref = _conv_ref(x, w, 1, 1, True, False, bias, case['act'], resid, scale=scale)
    assert lib.winograd_supported(cout, cin, H, W, prec) and not lib.winograd_supported(cout, cin, 3, W, prec) and not lib.winograd_supported(cout, cin, H, 6, prec)
    assert not lib.winograd_supported(cout, cin, H, W, L.PREC_F32) and not lib.winograd_supported(96, cin, H, W, prec)
    wp = lib.pack_winograd_weight(w, scale, prec)
    ws = torch.zeros(lib.winograd_workspace_bytes(B, cout, H, W) // 4)
    ybuf = torch.full((B, cout + 3, H, W), 7.0)
    flag = torch.zeros(1, dtype=torch.int32)
    lib.winograd_conv3x3(L.view(xbuf, 1, cin), wp, L.view(ybuf, 2, cout), B, ws, bias, case['act'], None if resid is None else L.view(resid),
                         precision=prec, range_flag=flag)
    y = ybuf[:, 2:2 + cout]
    tol = dict(atol=1.2e-3, rtol=4e-4) if prec == L.PREC_BF16X3 else dict(atol=3e-4, rtol=1e-4)   # sums of 4 inputs x transformed weights: 2x the direct kernel's bound
    assert torch.allclose(y, ref, **tol), float((y - ref).abs().max())
    assert float(ybuf[:, :2].min()) == 7.0 and float(ybuf[:, -1].max()) == 7.0 and int(flag) == 0
    wd = lib.pack_conv_weight(w, scale, precision=prec)
    yd = torch.zeros(B, cout, H, W)
    lib.conv2d(L.view(xbuf, 1, cin), wd, L.view(yd), B, 3, 1, 1, L.PAD_REFLECT, False, bias, case['act'], None if resid is None else L.view(resid),
               precision=prec)
    assert float((y - yd).abs().max()) < 2 * tol['atol']
    with pytest.raises(L.LamaError):                                         # workspace too small
        lib.winograd_conv3x3(L.view(xbuf, 1, cin), wp, L.view(ybuf, 2, cout), B, ws[:16], bias, case['act'], None, precision=prec)
    if case['cin'] == 32 and cout == 128:
        # round 4 (v108): the GEMM half alone (LAMA_CONV_DEFER_OUT), then the output transform on its own / inside an rfft2 launch -- the same bits
        for fused in (False, True):
            y2 = torch.full((B, cout + 3, H, W), 7.0)
            ws.zero_()
            pend = lib.winograd_conv3x3(L.view(xbuf, 1, cin), wp, L.view(y2, 2, cout), B, ws, bias, case['act'], None if resid is None else L.view(resid),
                                        precision=prec, defer_out=True)
            assert float(y2.min()) == 7.0 and pend is not None
            if fused:
                xf = torch.randn(1, 2, 64, 64, generator=g)
                sp, sp0 = torch.zeros(1, 4, 64, 33), torch.zeros(1, 4, 64, 33)
                lib.rfft2_wino_out(L.view(xf), L.view(sp), 1, None, pend, ws)
                lib.rfft2(L.view(xf), L.view(sp0), 1, None)
                assert torch.equal(sp, sp0)
                with pytest.raises(L.LamaError) as ei:                       # 32 x 32 planes: no kernel does both
                    lib.rfft2_wino_out(L.view(xf[:, :, :32, :32].contiguous()), L.view(torch.zeros(1, 4, 32, 17)), 1, None, pend, ws)
                assert ei.value.code == L.ERR_UNSUPPORTED
            else:
                lib.winograd_out(pend, ws)
            assert torch.equal(y2, ybuf)
    if case['cin'] <= 64 and cout == 128:
        # round 4: zero padding (the dgrad convs of the reverse pass): rows / columns outside the plane read as zeros, first / last band and column
        refz = _conv_ref(x, w, 1, 1, False, False, bias, case['act'], resid, scale=scale)
        lib.winograd_conv3x3(L.view(xbuf, 1, cin), wp, L.view(ybuf, 2, cout), B, ws, bias, case['act'], None if resid is None else L.view(resid),
                             precision=prec, range_flag=flag, pad_mode=L.PAD_ZERO)
        assert torch.allclose(ybuf[:, 2:2 + cout], refz, **tol), float((ybuf[:, 2:2 + cout] - refz).abs().max())
        assert float((refz - ref).abs().max()) > 0.1
