"""GPU parity tests (run with -m gpu on an MI355X): the HIP library through the C ABI against torch fp32
ops / the CPU oracle / the reference-generated golden vectors."""
import os

import numpy as np
import pytest
import torch

from lama_amd import _lib as L

pytestmark = pytest.mark.gpu

from tests.test_kernels_emu import (WINO_CASES, CONV_CASES, CONV_TOL, F16_CASES, F16_MIXED, F16_STEM_HEAD, FFT_SIZES, PREC_IDS, PRECISIONS, _conv_f16_ref, _conv_ref,  # noqa: E402
                                    _inv_ref, _spec_ref)


@pytest.fixture(scope='module')
def lib():
    """The PRODUCT library under its production kernel selection: liblama_hip.so reads no environment variable, so the
    LAMA_*=2 overrides tests/emu.py sets for the emulator (and for ``lib_forced`` below) do not reach it."""
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    return L.get_lib()          # raises if liblama_hip.so was not built: no fallback


@pytest.fixture(scope='module')
def lib_forced():
    """The -DLAMA_PROFILING build of the same sources with the specialised kernels forced at any launch size (LAMA_GEMM_WS /
    LAMA_CW_1X1 / LAMA_STEM_WS / LAMA_HEAD_WS = 2, set by tests/emu.py): the ragged-edge cases of CONV_CASES reach the
    persistent GEMM / stem / head / full-M kernels on hardware although production would not pick them at these sizes."""
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    from lama_amd import build as B
    assert os.environ.get('LAMA_GEMM_WS') == '2' and os.environ.get('LAMA_STEM_WS') == '2'
    return L.LamaLib(B.LIB_PROF)


@pytest.fixture(params=['production', 'forced'])
def anylib(request, lib, lib_forced):
    return lib if request.param == 'production' else lib_forced


DEV = 'cuda'


@pytest.mark.parametrize('prec', PRECISIONS, ids=PREC_IDS)
@pytest.mark.parametrize('case', CONV_CASES, ids=lambda c: f"k{c['k']}s{c['stride']}c{c['cin']}o{c['cout']}{'T' if c.get('transposed') else ''}{'w4s%d' % c['w4_slots'] if 'w4_slots' in c else ''}{'wks%d' % c['wk_slots'] if 'wk_slots' in c else ''}{'ct%d' % c['ct'] if 'ct' in c else ''}{'ctg%d' % c['ct_grid'] if 'ct_grid' in c else ''}")
def test_conv2d(anylib, case, prec, monkeypatch):
    lib = anylib
    if 'ct' in case:
        monkeypatch.setenv('LAMA_CT', str(case['ct']))
    if 'ct_grid' in case:
        monkeypatch.setenv('LAMA_CT_GRID', str(case['ct_grid']))
    if 'w4_slots' in case:  # pointwise GEMM over super-tiles of interleaved MFMA tiles, forced at these small sizes (profiling build switch)
        monkeypatch.setenv('LAMA_GEMM_W4', '2')
        monkeypatch.setenv('LAMA_GEMM_W4_SLOTS', str(case['w4_slots']))
    if 'wk_slots' in case:  # the spectral GEMM with all of K in one wave (gemm_wk_dev.inc), forced at these small sizes (profiling build switch)
        monkeypatch.setenv('LAMA_GEMM_WK', '2')
        monkeypatch.setenv('LAMA_GEMM_WK_SLOTS', str(case['wk_slots']))
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
    wp = lib.pack_conv_weight(w.to(DEV), None if scale is None else scale.to(DEV), stride=case['stride'], transposed=tr, precision=prec)
    ybuf = torch.full((B, cout + 3, ref.shape[2], ref.shape[3]), 7.0, device=DEV)
    xd = x.to(DEV)
    rd = None if resid is None else resid.to(DEV)
    bd = None if bias is None else bias.to(DEV)
    st = torch.cuda.current_stream().cuda_stream
    lib.conv2d(L.view(xd), wp, L.view(ybuf, 2, cout), B, k, case['stride'], case['pad'],
               L.PAD_ZERO if tr else L.PAD_REFLECT, tr, bd, case['act'], None if rd is None else L.view(rd), precision=prec, stream=st)
    torch.cuda.synchronize()
    y = ybuf[:, 2:2 + cout].cpu()
    assert torch.allclose(y, ref, **CONV_TOL[prec]), float((y - ref).abs().max())
    assert float(ybuf[:, :2].min()) == 7.0 and float(ybuf[:, -1].max()) == 7.0


@pytest.mark.parametrize('prec', PRECISIONS, ids=PREC_IDS)
@pytest.mark.parametrize('shape', [(2, 128, 128, 3, 64, 64), (2, 512, 128, 3, 64, 64), (1, 192, 384, 1, 64, 33), (1, 64, 3, 7, 96, 96),
                                   (1, 4, 64, 7, 128, 96), (1, 256, 128, 3, 40, 56), (1, 384, 192, 1, 64, 64),
                                   (4, 384, 192, 1, 64, 64), (6, 384, 384, 1, 64, 33),   # persistent pointwise GEMM (conv_ws_dev.inc), K = 384
                                   (8, 384, 384, 1, 64, 33), (2, 384, 384, 1, 128, 65), (4, 384, 384, 1, 128, 65),  # all of K in one wave (gemm_wk_dev.inc): 264 / 260 super-tiles = one round + the fifth waves; 520 go to the w4 kernel
                                   (8, 192, 192, 1, 64, 64),                             # ... K = 192
                                   (2, 4, 64, 7, 512, 256),                              # stem kernel (conv_stem_dev.inc)
                                   (2, 64, 3, 7, 512, 416)])                             # head kernel (conv_head_dev.inc)
def test_conv2d_big_tiles(lib, shape, prec):
    """Full-size channel counts (BM=128 tiles, many K chunks) against torch fp32 on the GPU's host."""
    B, cin, cout, k, H, W = shape
    g = torch.Generator().manual_seed(4)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    bias = torch.randn(cout, generator=g)
    ref = _conv_ref(x, w, 1, k // 2, True, False, bias, 1, None)
    y = torch.empty(B, cout, H, W, device=DEV)
    xd, bd = x.to(DEV), bias.to(DEV)             # keep the device copies alive until the kernel ran
    wp = lib.pack_conv_weight(w.to(DEV), None, precision=prec)
    lib.conv2d(L.view(xd), wp, L.view(y), B, k, 1, k // 2, L.PAD_REFLECT, False, bd, L.ACT_RELU, precision=prec,
               stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert torch.allclose(y.cpu(), ref, atol=3e-4, rtol=1e-4), float((y.cpu() - ref).abs().max())


@pytest.mark.parametrize('prec', [L.PREC_F16X3, L.PREC_BF16X3], ids=['f16x3', 'bf16x3'])
@pytest.mark.parametrize('shape', [(8, 512, 128, 64, 64), (7, 512, 128, 64, 64), (6, 512, 128, 64, 64), (10, 512, 128, 64, 64), (11, 512, 128, 64, 64),
                                   (16, 256, 128, 64, 64), (2, 64, 128, 9, 12)],
                         ids=['c2_256wg', '224wg_on', '192wg_off', '320wg_on', '352wg_off', '512wg_two_per_cu', 'small'])
def test_conv2d_local_conv_geometries(lib, shape, prec):
    """The local 3x3 conv's geometries by launch size (conv_wreg_host.inc): with LAMA_CONV_COOPERATIVE (lama_hip.h) one 4-wave workgroup
    per CU from 224 to 320 workgroups; two 4-wave workgroups per CU from 512 on; the 8-wave K-split workgroup otherwise.  Each against torch
    fp32, and the flagged launch against the plain one (different K splits: same value up to fp32 summation order)."""
    B, cin, cout, H, W = shape
    g = torch.Generator().manual_seed(14)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    bias, resid = torch.randn(cout, generator=g), torch.randn(B, cout, H, W, generator=g)
    ref = _conv_ref(x, w, 1, 1, True, False, bias, 1, resid)
    xd, bd, rd = x.to(DEV), bias.to(DEV), resid.to(DEV)
    wp = lib.pack_conv_weight(w.to(DEV), None, precision=prec)
    ys = []
    for coop in (False, True):
        y = torch.full((B, cout, H, W), 7.0, device=DEV)
        lib.conv2d(L.view(xd), wp, L.view(y), B, 3, 1, 1, L.PAD_REFLECT, False, bd, L.ACT_RELU, L.view(rd), precision=prec,
                   stream=torch.cuda.current_stream().cuda_stream, cooperative=coop)
        torch.cuda.synchronize()
        ys.append(y.cpu())
        assert torch.allclose(ys[-1], ref, atol=3e-4, rtol=1e-4), (coop, float((ys[-1] - ref).abs().max()))
    assert float((ys[0] - ys[1]).abs().max()) < (2e-4 if prec == L.PREC_BF16X3 else 2e-5)


# Shapes on BOTH sides of every size threshold of the production kernel selection (conv_wreg_host.inc): the specialised kernel
# just takes the launch / the launch just falls through to the next kernel in line.
#   gw_try_launch   (persistent pointwise GEMM): C in {192, 384}, rows % 96 == 0, 32-pixel tiles x row groups >= 64
#   cw_try_launch   (full-M 1x1 workgroups):     128-pixel tiles >= 96  (cin % 64 == 0, rows = 384 / 192, not taken by gw)
#   stem_try_launch (7x7, cin <= 4):             B * H * ceil(W / 32) >= 4096
#   head_try_launch (7x7, cin 64, cout <= 4):    B * ceil(W / 26) * H >= 8192
THRESHOLD_SHAPES = [
    # B, cin, cout, k, H, W
    (2, 192, 96, 1, 32, 32), (2, 192, 96, 1, 31, 32),          # gw: 64 tiles x 1 group on / 62 off (-> 128-row wreg tiles refused: generic)
    (1, 384, 192, 1, 32, 32), (1, 384, 192, 1, 32, 31),        # gw: 32 tiles x 2 groups on / 31 x 2 off (-> wreg 6 x 2 or generic)
    (3, 128, 384, 1, 64, 64), (2, 128, 384, 1, 64, 95),        # cw full-M 12 x 1: 3 * 32 = 96 tiles on, 2 * 48 = 96 (ragged) on
    (1, 128, 384, 1, 64, 95), (1, 256, 192, 1, 95, 128),       # cw: 48 tiles off (-> 128-row wreg tiles) / 95 tiles, 192 rows off (-> LDS-staged)
    (2, 4, 64, 7, 64, 1024), (2, 4, 64, 7, 63, 1024),          # stem: 4096 on / 4032 off (LDS-staged MODE 1 kernel)
    (2, 64, 3, 7, 256, 416), (2, 64, 3, 7, 255, 416),          # head: 8192 on / 8160 off (LDS-staged MODE 2 kernel)
]


@pytest.mark.parametrize('prec', [L.PREC_F16X3, L.PREC_BF16X3], ids=['f16x3', 'bf16x3'])
@pytest.mark.parametrize('shape', THRESHOLD_SHAPES, ids=lambda s: 'x'.join(map(str, s)))
def test_conv2d_selection_thresholds(lib, shape, prec):
    B, cin, cout, k, H, W = shape
    g = torch.Generator().manual_seed(44)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    bias = torch.randn(cout, generator=g)
    ref = _conv_ref(x, w, 1, k // 2, True, False, bias, 1, None)
    y = torch.empty(B, cout, H, W, device=DEV)
    xd, bd = x.to(DEV), bias.to(DEV)
    wp = lib.pack_conv_weight(w.to(DEV), None, precision=prec)
    lib.conv2d(L.view(xd), wp, L.view(y), B, k, 1, k // 2, L.PAD_REFLECT, False, bd, L.ACT_RELU, precision=prec,
               stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert torch.allclose(y.cpu(), ref, atol=3e-4, rtol=1e-4), float((y.cpu() - ref).abs().max())


@pytest.mark.parametrize('prec', PRECISIONS, ids=PREC_IDS)
@pytest.mark.parametrize('shape', [(2, 128, 64, 40, 72), (1, 256, 130, 24, 40), (2, 512, 256, 16, 16)], ids=lambda s: f'c{s[1]}o{s[2]}')
def test_conv_transpose_full_size(lib, shape, prec):
    """ConvTranspose2d(k3, s2, p1, op1) + folded BN + ReLU at the upsampling layers' channel counts (ffc.py:348-354): the split
    paths run it as ONE launch with four accumulator sets, the exact-fp32 path as four parity-class launches."""
    B, cin, cout, H, W = shape
    g = torch.Generator().manual_seed(21)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cin, cout, 3, 3, generator=g) / (cin * 2.25) ** 0.5
    scale, bias = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g)
    ref = _conv_ref(x, w, 2, 1, False, True, bias, 1, None, scale=scale)
    wp = lib.pack_conv_weight(w.to(DEV), scale.to(DEV), stride=2, transposed=True, precision=prec)
    y = torch.full((B, cout, 2 * H, 2 * W), 7.0, device=DEV)
    xd, bd = x.to(DEV), bias.to(DEV)
    lib.conv2d(L.view(xd), wp, L.view(y), B, 3, 2, 1, L.PAD_ZERO, True, bd, L.ACT_RELU, precision=prec, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    err = float((y.cpu() - ref).abs().max())
    assert err < 3e-4, err


@pytest.mark.parametrize('prec', PRECISIONS, ids=PREC_IDS)
def test_conv2d_fused_second_operand_full_size(lib, prec):
    """The bottleneck global-branch launch at full channel counts: relu(conv3x3(x_l) + conv1x1(t) + b) + resid."""
    g = torch.Generator().manual_seed(12)
    B, cl, cg, half, H, W = 2, 128, 384, 192, 64, 64
    state = torch.randn(B, cl + cg, H, W, generator=g)
    t = torch.randn(B, half, H, W, generator=g)
    w1 = torch.randn(cg, cl, 3, 3, generator=g) / (cl * 9) ** 0.5
    w2 = torch.randn(cg, half, 1, 1, generator=g) / half ** 0.5
    scale, bias = torch.rand(cg, generator=g) + 0.5, torch.randn(cg, generator=g)
    ref = _conv_ref(state[:, :cl], w1, 1, 1, True, False, bias, 1, state[:, cl:], x2=t, w2=w2 * scale[:, None, None, None], scale=scale)
    sd, td, bd, scd = state.to(DEV), t.to(DEV), bias.to(DEV), scale.to(DEV)
    out = torch.zeros_like(sd)
    wp1, wp2 = lib.pack_conv_weight(w1.to(DEV), scd, precision=prec), lib.pack_conv_weight(w2.to(DEV), scd, precision=prec)
    lib.conv2d(L.view(sd, 0, cl), wp1, L.view(out, cl, cg), B, 3, 1, 1, L.PAD_REFLECT, False, bd, L.ACT_RELU, L.view(sd, cl, cg),
               x2=L.view(td), w2_packed=wp2, precision=prec, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    err = float((out[:, cl:].cpu() - ref).abs().max())
    assert err < 3e-4, err


@pytest.mark.parametrize('hw', FFT_SIZES + [(128, 128), (512, 256), (168, 168), (135, 240), (125, 188), (96, 128), (127, 131), (128, 256), (160, 120), (199, 100), (192, 256), (251, 64), (270, 480), (375, 500), (188, 376)], ids=lambda s: f'{s[0]}x{s[1]}')
def test_rfft2_irfft2(lib, hw):
    h, w = hw
    g = torch.Generator().manual_seed(h * 131 + w)
    B, Cn = 2, 3
    wide = torch.randn(B, Cn + 2, h, w, generator=g)
    x = wide[:, 1:1 + Cn]
    wd = wide.to(DEV)
    spec = torch.zeros(B, 2 * Cn, h, w // 2 + 1, device=DEV)
    ws = torch.zeros(max(lib.fft_workspace_bytes(B, Cn, h, w), 4) // 4, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    lib.rfft2(L.view(wd, 1, Cn), L.view(spec), B, ws, stream=st)
    ref = _spec_ref(x)
    tol = 3e-5 if max(h, w) <= 128 else 2e-4
    assert torch.allclose(spec.cpu(), ref, atol=tol, rtol=1e-4), float((spec.cpu() - ref).abs().max())
    spec2 = torch.relu(torch.randn(B, 2 * Cn, h, w // 2 + 1, generator=g))
    resid = torch.randn(B, Cn, h, w, generator=g)
    y = torch.zeros(B, Cn, h, w, device=DEV)
    s2d, rd = spec2.to(DEV), resid.to(DEV)
    lib.irfft2(L.view(s2d), L.view(rd), L.view(y), B, ws, stream=st)
    ref2 = resid + _inv_ref(spec2, h, w)
    assert torch.allclose(y.cpu(), ref2, atol=tol, rtol=1e-4), float((y.cpu() - ref2).abs().max())


def test_rfft2_irfft2_every_length_up_to_256(lib):
    """Every plane length the /8 bottleneck of a padded input can have (evaluation/data.py:29-33 pads to multiples of 8 only): h = 2 .. 256 paired with
    a w that walks the same range in another order -- every mixed-radix plan (composite radices, 11, 13, the pair-symmetric pass of larger primes, the
    two-launch form beyond the LDS) as rows AND as columns, against torch.fft on the host."""
    st = torch.cuda.current_stream().cuda_stream
    B, Cn = 1, 2
    worst = (0.0, None)
    for h in range(2, 257):
        w = 2 + (h * 97) % 255                                   # 97 is coprime to 255: w takes every value of 2 .. 256 once
        g = torch.Generator().manual_seed(h)
        x = torch.randn(B, Cn, h, w, generator=g)
        xd = x.to(DEV)
        spec = torch.zeros(B, 2 * Cn, h, w // 2 + 1, device=DEV)
        ws = torch.zeros(max(lib.fft_workspace_bytes(B, Cn, h, w), 4) // 4, device=DEV)
        lib.rfft2(L.view(xd), L.view(spec), B, ws, stream=st)
        ref = _spec_ref(x)
        e1 = float((spec.cpu() - ref).abs().max())
        spec2 = torch.relu(torch.randn(B, 2 * Cn, h, w // 2 + 1, generator=g))
        resid = torch.randn(B, Cn, h, w, generator=g)
        y = torch.zeros(B, Cn, h, w, device=DEV)
        s2d, rd = spec2.to(DEV), resid.to(DEV)                    # (kept alive: L.view holds raw pointers)
        lib.irfft2(L.view(s2d), L.view(rd), L.view(y), B, ws, stream=st)
        e2 = float((y.cpu() - (resid + _inv_ref(spec2, h, w))).abs().max())
        if max(e1, e2) > worst[0]:
            worst = (max(e1, e2), (h, w))
        assert e1 < 2e-4 and e2 < 2e-4, (h, w, e1, e2)
    print('worst', worst)


def test_winograd_rows16_staging_is_bit_identical(lib_forced, monkeypatch):
    """GEO 1 of wino_gemm_kernel (64-column planes: halo columns by row DPP inside the 16-lane row that is the tile row, halo differences from
    v_sub_f32_dpp -- round 6, fourth session) against the any-geometry staging of the same kernel (LAMA_WG_GEO=0 in the profiling build): the same
    arithmetic on the same values in the same order, so the partial sums and the output are equal bit for bit -- edge tiles (reflected columns 1 and
    W - 2) included, for the bottleneck's shape and a ragged batch."""
    lib = lib_forced
    st = torch.cuda.current_stream().cuda_stream
    for B, cin, cout, H in ((8, 512, 128, 64), (3, 64, 256, 16)):
        g = torch.Generator().manual_seed(B * 100 + H)
        x = torch.randn(B, cin, H, 64, generator=g).to(DEV)
        w = (torch.randn(cout, cin, 3, 3, generator=g) * 0.03).to(DEV)
        bias, resid = torch.randn(cout, generator=g).to(DEV), torch.randn(B, cout, H, 64, generator=g).to(DEV)
        wp = lib.pack_winograd_weight(w, None, L.PREC_F16X3)
        outs = []
        for geo in ('0', '1'):
            monkeypatch.setenv('LAMA_WG_GEO', geo)
            ws = torch.zeros(lib.winograd_workspace_bytes(B, cout, H, 64) // 4, device=DEV)
            y = torch.zeros(B, cout, H, 64, device=DEV)
            lib.winograd_conv3x3(L.view(x), wp, L.view(y), B, ws, bias, L.ACT_RELU, L.view(resid), precision=L.PREC_F16X3, stream=st)
            torch.cuda.synchronize()
            outs.append((y.clone(), ws.clone()))
        assert torch.equal(outs[0][1], outs[1][1]), 'partial sums differ'
        assert torch.equal(outs[0][0], outs[1][0])
        assert float(outs[1][0].abs().max()) > 0


def test_fft_masked_entries(lib):
    """lama_rfft2_masked_fwd / lama_irfft2_masked_fwd (v108) on 256 x 256 planes against transform + separate mask, channel views of wider buffers."""
    g = torch.Generator().manual_seed(77)
    B, Cn, h, w = 2, 192, 256, 256
    wide = torch.randn(B, Cn + 2, h, w, generator=g).to(DEV)
    ws = torch.zeros(max(lib.fft_workspace_bytes(B, Cn, h, w), 4) // 4, device=DEV)
    ms = torch.randn(B, 2 * Cn, h, w // 2 + 1, generator=g).to(DEV)
    spec, spec_m = torch.zeros(B, 2 * Cn, h, w // 2 + 1, device=DEV), torch.full((B, 2 * Cn, h, w // 2 + 1), 3.0, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    lib.rfft2(L.view(wide, 1, Cn), L.view(spec), B, ws, stream=st)
    lib.rfft2(L.view(wide, 1, Cn), L.view(spec_m), B, ws, stream=st, mask=L.view(ms))
    assert torch.equal(spec_m, spec * (ms > 0))
    spec2 = torch.relu(torch.randn(B, 2 * Cn, h, w // 2 + 1, generator=g)).to(DEV)
    resid, my = torch.randn(B, Cn, h, w, generator=g).to(DEV), torch.randn(B, Cn + 1, h, w, generator=g).to(DEV)
    y, y_m = torch.zeros(B, Cn, h, w, device=DEV), torch.full((B, Cn, h, w), 3.0, device=DEV)
    lib.irfft2(L.view(spec2), L.view(resid), L.view(y), B, ws, stream=st)
    lib.irfft2(L.view(spec2), L.view(resid), L.view(y_m), B, ws, stream=st, mask=L.view(my, 1, Cn))
    assert torch.equal(y_m, y * (my[:, 1:] > 0))
    with pytest.raises(L.LamaError) as ei:
        lib.rfft2(L.view(wide[:, :, :64, :64].contiguous(), 0, 4), L.view(torch.zeros(B, 8, 64, 33, device=DEV)), B, ws, stream=st,
                  mask=L.view(torch.ones(B, 8, 64, 33, device=DEV)))
    assert ei.value.code == L.ERR_UNSUPPORTED


@pytest.mark.parametrize('n_seq', [(64, 1), (64, 2), (64, 3), (128, 2), (128, 3)], ids=lambda s: f'{s[0]}seq{s[1]}')
def test_fft_sequential_planes(lib_forced, n_seq, monkeypatch):
    lib = lib_forced      # LAMA_FFT_SEQ / LAMA_FFT_INPLACE exist in the profiling build only
    """Sized one-plane FFT kernels walking LAMA_FFT_SEQ consecutive planes per workgroup (next plane prefetched)."""
    n, seq = n_seq
    monkeypatch.setenv('LAMA_FFT_SEQ', str(seq))
    monkeypatch.setenv('LAMA_FFT_INPLACE', '0')     # the one-buffer 64 x 64 kernels (default) are covered by test_rfft2_irfft2*
    g = torch.Generator().manual_seed(n + seq)
    B, Cn = 2, 12
    x = torch.randn(B, Cn, n, n, generator=g)
    xd = x.to(DEV)
    spec = torch.zeros(B, 2 * Cn, n, n // 2 + 1, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    lib.rfft2(L.view(xd), L.view(spec), B, None, stream=st)
    ref = _spec_ref(x)
    assert torch.allclose(spec.cpu(), ref, atol=3e-5, rtol=1e-4), float((spec.cpu() - ref).abs().max())
    spec2 = torch.relu(torch.randn(B, 2 * Cn, n, n // 2 + 1, generator=g))
    resid = torch.randn(B, Cn, n, n, generator=g)
    y = torch.zeros(B, Cn, n, n, device=DEV)
    s2d, rd = spec2.to(DEV), resid.to(DEV)
    lib.irfft2(L.view(s2d), L.view(rd), L.view(y), B, None, stream=st)
    ref2 = resid + _inv_ref(spec2, n, n)
    assert torch.allclose(y.cpu(), ref2, atol=3e-5, rtol=1e-4), float((y.cpu() - ref2).abs().max())


@pytest.mark.parametrize('prec', PRECISIONS, ids=PREC_IDS)
def test_fourier_unit_c2_shape(lib, prec):
    """FourierUnit at the BASELINE config-2 shape [8,192,64,64] against the oracle."""
    from oracle import lama_oracle as O
    g = torch.Generator().manual_seed(5)
    B, Cn, h, w = 8, 192, 64, 64
    x = torch.randn(B, Cn, h, w, generator=g)
    sd = {'fu.conv_layer.weight': torch.randn(2 * Cn, 2 * Cn, 1, 1, generator=g) / (2 * Cn) ** 0.5,
          'fu.bn.weight': torch.rand(2 * Cn, generator=g) + 0.5, 'fu.bn.bias': torch.randn(2 * Cn, generator=g) * 0.2,
          'fu.bn.running_mean': torch.randn(2 * Cn, generator=g) * 0.1, 'fu.bn.running_var': torch.rand(2 * Cn, generator=g) + 0.5}
    with torch.no_grad():
        ref = x + O.fourier_unit(x, sd, 'fu')
    scale = sd['fu.bn.weight'] / torch.sqrt(sd['fu.bn.running_var'] + 1e-5)
    shift = sd['fu.bn.bias'] - sd['fu.bn.running_mean'] * scale
    wp = lib.pack_conv_weight(sd['fu.conv_layer.weight'].to(DEV), scale.to(DEV), precision=prec)
    ws = torch.zeros(lib.fourier_unit_workspace_bytes(B, Cn, h, w) // 4 + 1, device=DEV)
    xd = x.to(DEV)
    y = torch.zeros_like(xd)
    shd = shift.to(DEV)
    lib.fourier_unit(L.view(xd), wp, shd, L.view(y), B, True, ws, precision=prec, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert float((y.cpu() - ref).abs().max()) < 1e-4


@pytest.mark.parametrize('shape', [(8, 192, 64, 64), (4, 192, 128, 128), (1, 192, 256, 256), (1, 48, 135, 240)],
                         ids=['c2_64', 'c3_128', 'c5_256', 'photo_135x240'])
def test_fft_round_trip_and_parseval_full_size(lib, shape):
    """Size-independent properties at the BASELINE plane sizes (no CPU reference needed): irfft2(rfft2(x)) = x, the fused residual
    doubles it (x + irfft2(rfft2(x)) = 2x), and the ortho transform keeps the energy (Parseval on the half spectrum with the
    interior kx bins counted twice)."""
    B, Cn, h, w = shape
    g = torch.Generator().manual_seed(h + w)
    x = torch.randn(B, Cn, h, w, generator=g).to(DEV)
    wf = w // 2 + 1
    spec = torch.zeros(B, 2 * Cn, h, wf, device=DEV)
    ws = torch.zeros(max(lib.fft_workspace_bytes(B, Cn, h, w), 4) // 4, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    lib.rfft2(L.view(x), L.view(spec), B, ws, stream=st)
    y = torch.zeros_like(x)
    lib.irfft2(L.view(spec), None, L.view(y), B, ws, stream=st)
    y2 = torch.zeros_like(x)
    lib.irfft2(L.view(spec), L.view(x), L.view(y2), B, ws, stream=st)
    torch.cuda.synchronize()
    tol = 2e-5 if max(h, w) <= 128 else 1e-4
    assert float((y - x).abs().max()) < tol and float((y2 - 2 * x).abs().max()) < 2 * tol
    sp = spec.view(B, Cn, 2, h, wf).double()
    wgt = torch.full((wf,), 2.0, dtype=torch.float64, device=DEV)
    wgt[0] = 1.0
    if w % 2 == 0:
        wgt[-1] = 1.0
    e_spec = ((sp ** 2).sum(2) * wgt).sum(dim=(-2, -1))
    e_x = (x.double() ** 2).sum(dim=(-2, -1))
    assert float(((e_spec - e_x).abs() / e_x).max()) < 1e-5


@pytest.mark.parametrize('prec', [L.PREC_F16X3, L.PREC_BF16X3], ids=['f16x3', 'bf16x3'])
def test_conv_linearity_full_size(lib, prec):
    """conv(a x + b y) = a conv(x) + b conv(y) for the local 3x3 launch at the BASELINE configs[1] shape [8,512,64,64] -> 128 channels
    (no bias / activation): a property of the launch that needs no CPU reference at this size."""
    g = torch.Generator().manual_seed(17)
    B, cin, cout, H, W = 8, 512, 128, 64, 64
    x, y = torch.randn(B, cin, H, W, generator=g).to(DEV), torch.randn(B, cin, H, W, generator=g).to(DEV)
    wt = (torch.randn(cout, cin, 3, 3, generator=g) * 0.02).to(DEV)
    wp = lib.pack_conv_weight(wt, None, precision=prec)
    st = torch.cuda.current_stream().cuda_stream

    def conv(t):
        o = torch.zeros(B, cout, H, W, device=DEV)
        lib.conv2d(L.view(t), wp, L.view(o), B, 3, 1, 1, L.PAD_REFLECT, False, None, L.ACT_NONE, None, precision=prec, stream=st)
        return o
    a, b = 0.75, -1.5
    lhs, rhs = conv(a * x + b * y), a * conv(x) + b * conv(y)
    torch.cuda.synchronize()
    scale = float(rhs.abs().max())
    assert float((lhs - rhs).abs().max()) < (2e-5 if prec == L.PREC_F16X3 else 2e-4) * max(1.0, scale)


@pytest.mark.parametrize('prec', [L.PREC_F16X3, L.PREC_BF16X3], ids=['f16x3', 'bf16x3'])
@pytest.mark.parametrize('hw', [(64, 64), (9, 37)], ids=['c2_64x64', 'ragged_9x37'])
def test_conv2d_fused_next_conv1(lib, lib_forced, prec, hw):
    """lama_conv2d_args.fuse1_*: SpectralTransform.conv1 of the next layer in the epilogue of the global-branch launch (3x3 over x_l +
    1x1 over t + bias + ReLU + residual -> 384 channels).  y is bit-identical to the plain launch; x1 equals conv1 run on y as a launch
    of its own (same products, another summation order) and the fp64 reference.  The ragged case is below the 160 tiles from which
    production takes the 12-wave all-rows workgroup (the only one that carries conv1; smaller launches answer LAMA_ERR_UNSUPPORTED and
    FFC.launch keeps conv1 a launch of its own): it runs on the profiling build with LAMA_CW_G12=2 (tests/emu.py)."""
    if hw[0] != 64:
        assert os.environ.get('LAMA_CW_G12') == '2'
        lib = lib_forced
    g = torch.Generator().manual_seed(23)
    B, cl, cg, half, (H, W) = (8 if hw[0] == 64 else 2), 128, 384, 192, hw
    xl, t = torch.randn(B, cl, H, W, generator=g), torch.randn(B, half, H, W, generator=g)
    w1, w2 = torch.randn(cg, cl, 3, 3, generator=g) * 0.03, torch.randn(cg, half, 1, 1, generator=g) * 0.05
    scale, bias = torch.rand(cg, generator=g) + 0.5, torch.randn(cg, generator=g)
    resid = torch.randn(B, cg, H, W, generator=g)
    wc1 = torch.randn(192, cg, 1, 1, generator=g) * 0.05
    s1, b1 = torch.rand(192, generator=g) + 0.5, torch.randn(192, generator=g) * 0.2
    st = torch.cuda.current_stream().cuda_stream
    d = lambda v: v.to(DEV)
    wp1, wp2 = lib.pack_conv_weight(d(w1), d(scale), precision=prec), lib.pack_conv_weight(d(w2), d(scale), precision=prec)
    order = lib.fuse1_channel_order()
    assert sorted(order.tolist()) == list(range(384))
    wpc_f = lib.pack_conv_weight(d(wc1[:, order].contiguous()), d(s1), precision=prec)
    wpc = lib.pack_conv_weight(d(wc1), d(s1), precision=prec)
    xld, td, rd, bd, b1d = d(xl), d(t), d(resid), d(bias), d(b1)
    y0 = torch.zeros(B, cg, H, W, device=DEV); y = torch.zeros_like(y0)
    x1 = torch.zeros(B, 192, H, W, device=DEV); x1s = torch.zeros_like(x1)
    common = dict(x2=L.view(td), w2_packed=wp2, precision=prec, stream=st)
    lib.conv2d(L.view(xld), wp1, L.view(y0), B, 3, 1, 1, L.PAD_REFLECT, False, bd, L.ACT_RELU, L.view(rd), **common)
    lib.conv2d(L.view(y0), wpc, L.view(x1s), B, 1, bias=b1d, act=L.ACT_RELU, precision=prec, stream=st)
    lib.conv2d(L.view(xld), wp1, L.view(y), B, 3, 1, 1, L.PAD_REFLECT, False, bd, L.ACT_RELU, L.view(rd), fuse1=(wpc_f, b1d, L.view(x1)), **common)
    torch.cuda.synchronize()
    assert torch.equal(y, y0)
    ref = torch.relu(torch.nn.functional.conv2d(y0.double().cpu(), (wc1 * s1[:, None, None, None]).double()) + b1.double()[None, :, None, None]).float()
    tol = (2e-5 if prec == L.PREC_F16X3 else 2e-4) * max(1.0, float(ref.abs().max()))
    assert float((x1.cpu() - ref).abs().max()) < tol and float((x1 - x1s).abs().max()) < tol
    with pytest.raises(L.LamaError):      # the fused stage exists for the 384-channel global-branch launch only
        lib.conv2d(L.view(xld), wp1, L.view(y), B, 3, 1, 1, L.PAD_REFLECT, False, bd, L.ACT_RELU, L.view(rd), fuse1=(wpc_f, b1d, L.view(x1)),
                   precision=prec, stream=st)


def test_elementwise(lib):
    g = torch.Generator().manual_seed(9)
    B, H, W = 2, 40, 56
    img, mask = torch.rand(B, 3, H, W, generator=g), (torch.rand(B, 1, H, W, generator=g) > 0.5).float()
    pred = torch.rand(B, 3, H, W, generator=g)
    st = torch.cuda.current_stream().cuda_stream
    out = torch.zeros(B, 4, H, W, device=DEV)
    imgd, maskd, predd = img.to(DEV), mask.to(DEV), pred.to(DEV)
    lib.mask_compose(L.view(imgd), L.view(maskd), L.view(out), B, st)
    assert torch.equal(out.cpu(), torch.cat([img * (1 - mask), mask], 1))
    bl = torch.zeros(B, 3, H, W, device=DEV)
    lib.blend(L.view(imgd), L.view(maskd), L.view(predd), L.view(bl), B, st)
    assert torch.allclose(bl.cpu(), mask * pred + (1 - mask) * img, atol=1e-6)
    src = torch.rand(B, 3, H, W, generator=g) * 1.2 - 0.1
    u8 = torch.zeros(B, 37, 50, 3, dtype=torch.uint8, device=DEV)
    srcd = src.to(DEV)
    lib.quantize_u8_hwc(L.view(srcd), u8, B, 37, 50, st)
    ref = np.clip(src.permute(0, 2, 3, 1).numpy()[:, :37, :50] * 255, 0, 255).astype('uint8')
    assert np.array_equal(u8.cpu().numpy(), ref)


def test_f16_split_range_watch(lib):
    """the default f16 split must not return inf / garbage silently.  A FourierUnit input whose DC bin
    leaves the fp16 range (|DC| = sqrt(h*w) * mean = 64 * 2000 > 65504) raises lama_fourier_unit_fwd's range flag on the f16
    split; the bf16 split takes the same input (flag untouched) and matches the oracle to its 16 mantissa bits."""
    from oracle import lama_oracle as O
    g = torch.Generator().manual_seed(6)
    B, Cn, h, w = 2, 192, 64, 64
    x = torch.randn(B, Cn, h, w, generator=g)
    sd = {'fu.conv_layer.weight': torch.randn(2 * Cn, 2 * Cn, 1, 1, generator=g) / (2 * Cn) ** 0.5,
          'fu.bn.weight': torch.rand(2 * Cn, generator=g) + 0.5, 'fu.bn.bias': torch.randn(2 * Cn, generator=g) * 0.2,
          'fu.bn.running_mean': torch.randn(2 * Cn, generator=g) * 0.1, 'fu.bn.running_var': torch.rand(2 * Cn, generator=g) + 0.5}
    scale = sd['fu.bn.weight'] / torch.sqrt(sd['fu.bn.running_var'] + 1e-5)
    shift = (sd['fu.bn.bias'] - sd['fu.bn.running_mean'] * scale).to(DEV)
    ws = torch.zeros(lib.fourier_unit_workspace_bytes(B, Cn, h, w) // 4 + 1, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    xbig = x.clone()
    xbig[1, 7] += 2000.0                                    # DC of that plane = 64 * 2000 = 128000
    for prec in (L.PREC_F16X3, L.PREC_BF16X3):
        wp = lib.pack_conv_weight(sd['fu.conv_layer.weight'].to(DEV), scale.to(DEV), precision=prec)
        for inp, overflow in ((x, False), (xbig, True)):
            flag = torch.zeros(1, dtype=torch.int32, device=DEV)
            xd = inp.to(DEV)
            y = torch.zeros_like(xd)
            lib.fourier_unit(L.view(xd), wp, shift, L.view(y), B, True, ws, precision=prec, stream=st, range_flag=flag)
            torch.cuda.synchronize()
            assert int(flag.item()) == (1 if (overflow and prec == L.PREC_F16X3) else 0), (prec, overflow)
            if not (overflow and prec == L.PREC_F16X3):
                with torch.no_grad():
                    ref = inp + O.fourier_unit(inp, sd, 'fu')
                tol = 1e-4 if not overflow else 0.5          # 128000 * 2^-17 per product on the bf16 split
                assert float((y.cpu() - ref).abs().max()) < tol


def test_fft_next_to_mfma_load_is_bit_identical(lib_forced):
    """Regression test of the round-1 co-residency hazard.  On MI355X a packed-fp32 VALU instruction with an op_sel
    swizzle returns wrong results while another kernel's MFMA instructions execute on the same SIMD: the FFT kernels (float2
    butterflies, SLP-vectorised into v_pk_*_f32 by hipcc) came out wrong in up to 100 % of the runs next to a convolution or a bare
    MFMA loop on a second stream.  The library is built without packed-fp32 instructions; here the FFT kernels run next to the MFMA
    spinner of the profiling build and next to the bottleneck 3x3 conv and must reproduce the serial result bit for bit.  The
    one-instruction probe (which keeps its v_pk_add_f32 op_sel on purpose) documents that the hardware behaviour is still there."""
    import ctypes as C
    lib = lib_forced
    hog, probe = lib._l.lama_debug_hog, lib._l.lama_debug_probe
    hog.restype, hog.argtypes = C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
    probe.restype, probe.argtypes = C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    g = torch.Generator().manual_seed(2)
    B, Cn, n = 8, 192, 64
    x1 = torch.randn(B, Cn, n, n, generator=g).to(DEV)
    s2 = torch.relu(torch.randn(B, 2 * Cn, n, n // 2 + 1, generator=g)).to(DEV)
    s1, t = torch.empty_like(s2), torch.empty_like(x1)
    hout = torch.empty(256 * 512, device=DEV)
    xa = torch.randn(B, 512, n, n, generator=g).to(DEV)
    ya = torch.empty(B, 128, n, n, device=DEV)
    wa = lib.pack_conv_weight((torch.randn(128, 512, 3, 3, generator=g) * 0.02).to(DEV), None, precision=L.PREC_F16X3)
    main, side = torch.cuda.current_stream(), torch.cuda.Stream()

    def ffts(s):
        lib.rfft2(L.view(x1), L.view(s1), B, None, s)
        lib.irfft2(L.view(s2), L.view(x1), L.view(t), B, None, s)

    ffts(main.cuda_stream)
    torch.cuda.synchronize()
    r1, rt = s1.clone(), t.clone()
    for load in ('mfma', 'conv'):
        wrong = 0
        for it in range(150):
            s1.fill_(7.0); t.fill_(7.0)
            side.wait_stream(main)
            if load == 'mfma':
                lib.check(hog(main.cuda_stream, 256, 3, 1200, hout.data_ptr()), 'hog')
            else:
                lib.conv2d(L.view(xa), wa, L.view(ya), B, 3, 1, 1, L.PAD_REFLECT, False, None, L.ACT_RELU, precision=L.PREC_F16X3, stream=main.cuda_stream)
            ffts(side.cuda_stream)
            main.wait_stream(side)
            torch.cuda.synchronize()
            wrong += int(not (torch.equal(s1, r1) and torch.equal(t, rt)))
        assert wrong == 0, (load, wrong)
    # the erratum itself: v_pk_add_f32 with op_sel:[0,1] op_sel_hi:[1,0] (probe mode 9), alone vs next to the MFMA spinner
    G = 1536
    src = torch.randn(G * 4096, generator=g).to(DEV)
    out = torch.empty_like(src)
    lib.check(probe(main.cuda_stream, G, 9, src.data_ptr(), out.data_ptr()), 'probe')
    torch.cuda.synchronize()
    ref = out.clone()
    hit = 0
    for it in range(20):
        out.fill_(7.0)
        side.wait_stream(main)
        lib.check(hog(main.cuda_stream, 256, 3, 1200, hout.data_ptr()), 'hog')
        lib.check(probe(side.cuda_stream, G, 9, src.data_ptr(), out.data_ptr()), 'probe')
        main.wait_stream(side)
        torch.cuda.synchronize()
        hit += int(not torch.equal(out, ref))
    print(f'v_pk_add_f32 op_sel probe next to an MFMA loop: wrong in {hit} of 20 runs (hardware behaviour, informational)')


def test_overlap_streams_bit_identical():
    """generator.overlap_streams (default on since round 2): the overlapped forward equals the serial one bit for bit, eager and
    from the captured hipGraph."""
    from lama_amd.modules import make_generator
    from oracle import lama_oracle as O
    cfg = O.BIG_LAMA
    gen = make_generator(None, kind='ffc_resnet', **cfg)
    gen.load_state_dict(O.make_synthetic_state_dict(cfg, seed=0, calib_hw=64), strict=True)
    gen.cuda()
    batch = O.make_synthetic_batch(4, 256, 256, seed=12)
    x = torch.cat([batch['image'] * (1 - batch['mask']), batch['mask']], 1).cuda()
    assert gen.overlap_streams
    gen.overlap_streams = False
    plain = gen(x)
    ex = next(m for m in gen.modules() if hasattr(m, '_exec'))._exec
    ex.cooperative_serial = True       # the overlapped order launches the local conv with LAMA_CONV_COOPERATIVE: same geometry here
    gen._plans.clear()
    ref = gen(x)
    ex.cooperative_serial = False
    assert float((ref - plain).abs().max()) < 1e-4     # the two geometries differ in fp32 summation order only
    gen._plans.clear()
    gen.overlap_streams = True
    for pipelined in (True, False):        # the local convs as a chain of their own (ffc.SidePipe) / a fork + join around every local conv
        gen.pipeline_local = pipelined
        for graph in (False, True):
            gen.use_graph = graph
            gen._plans.clear()
            for _ in range(25):
                assert torch.equal(gen(x), ref), (pipelined, graph)


def _run_f16_case_gpu(lib, case, x_dtype=torch.float16, y_dtype=torch.float16, B=2):
    g = torch.Generator().manual_seed(3)
    cin, cout, k = case['cin'], case['cout'], case['k']
    tr = case.get('transposed', False)
    x = torch.randn(B, cin, case['H'], case['W'], generator=g).to(x_dtype)
    w = torch.randn((cin, cout, k, k) if tr else (cout, cin, k, k), generator=g) * (case.get('wscale') or 0.2)
    scale = torch.rand(cout, generator=g) + 0.5 if case['scale'] else None
    bias = torch.randn(cout, generator=g) if case['bias'] else None
    ref0 = _conv_f16_ref(x.half(), w, case['stride'], case['pad'], True, tr, None, 0, None, scale=scale)
    resid = torch.randn(ref0.shape, generator=g).to(y_dtype) if case['resid'] else None
    ref = _conv_f16_ref(x.half(), w, case['stride'], case['pad'], True, tr, bias, case['act'], resid, scale=scale)
    wp = lib.pack_conv_weight(w.to(DEV), None if scale is None else scale.to(DEV), stride=case['stride'], transposed=tr, precision=L.PREC_F16)
    ybuf = torch.full((B, cout + 3, ref.shape[2], ref.shape[3]), 7.0, dtype=y_dtype, device=DEV)
    xd = x.to(DEV)
    rd = None if resid is None else resid.to(DEV)
    bd = None if bias is None else bias.to(DEV)
    lib.conv2d(L.view(xd), wp, L.view(ybuf, 2, cout), B, k, case['stride'], case['pad'], L.PAD_ZERO if tr else L.PAD_REFLECT, tr, bd,
               case['act'], None if rd is None else L.view(rd), precision=L.PREC_F16, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    y = ybuf[:, 2:2 + cout].float().cpu()
    tol = 2e-3 * max(1.0, float(ref.abs().max())) if y_dtype == torch.float16 else 2e-4 * max(1.0, float(ref.abs().max()))
    assert float((y - ref).abs().max()) < tol, float((y - ref).abs().max())
    assert float(ybuf[:, :2].float().min()) == 7.0 and float(ybuf[:, -1].float().max()) == 7.0


@pytest.mark.parametrize('case', F16_CASES, ids=lambda c: f"k{c['k']}s{c['stride']}c{c['cin']}o{c['cout']}{'T' if c.get('transposed') else ''}")
def test_conv2d_fp16_io(anylib, case):
    """LAMA_PREC_F16 (BASELINE configs[2]): fp16 tensors, one product per MAC -- every kernel family, production selection and forced."""
    _run_f16_case_gpu(anylib, case)


@pytest.mark.parametrize('case,xdt,ydt', F16_STEM_HEAD, ids=['stem_f32_to_f16', 'head_f16_to_f32'])
def test_conv2d_fp16_stem_head(anylib, case, xdt, ydt):
    _run_f16_case_gpu(anylib, case, xdt, ydt)


MIXED_IO = [(torch.float32, torch.float16), (torch.float16, torch.float32)]


@pytest.mark.parametrize('xdt,ydt', MIXED_IO, ids=['f32_to_f16', 'f16_to_f32'])
@pytest.mark.parametrize('case', F16_MIXED, ids=lambda c: f"k{c['k']}s{c['stride']}c{c['cin']}o{c['cout']}{'T' if c.get('transposed') else ''}")
def test_conv2d_fp16_mixed_io(anylib, case, xdt, ydt):
    """LAMA_PREC_F16 launches next to the fp32 residual stream of the resnet blocks."""
    _run_f16_case_gpu(anylib, case, xdt, ydt)


@pytest.mark.parametrize('sdt,odt', [(torch.float16, torch.float16)] + MIXED_IO, ids=['f16', 'state_f32', 'out_f32'])
@pytest.mark.parametrize('hw', [(6, 35), (64, 64)], ids=['6x35', '64x64'])
def test_conv2d_fp16_fused_second_operand(anylib, hw, sdt, odt):
    """The bottleneck global-branch launch at LAMA_PREC_F16 in the three element-type mixes the generator plan uses."""
    g = torch.Generator().manual_seed(3)
    B, cl, cg, half, (H, W) = 2, 128, 384, 192, hw
    state = torch.randn(B, cl + cg, H, W, generator=g).to(sdt)
    t = torch.randn(B, half, H, W, generator=g).half()
    w1, w2 = torch.randn(cg, cl, 3, 3, generator=g) * 0.03, torch.randn(cg, half, 1, 1, generator=g) * 0.05
    scale, bias = torch.rand(cg, generator=g) + 0.5, torch.randn(cg, generator=g)
    resid = torch.randn(B, cg, H, W, generator=g).to(odt)
    ref = _conv_f16_ref(state[:, :cl].half(), w1, 1, 1, True, False, bias, 1, resid, x2=t, w2=w2, scale=scale)
    out = torch.zeros(B, cl + cg, H, W, dtype=odt, device=DEV)
    sd_, td, rd, sc = state.to(DEV), t.to(DEV), resid.to(DEV), scale.to(DEV)
    anylib.conv2d(L.view(sd_, 0, cl), anylib.pack_conv_weight(w1.to(DEV), sc, precision=L.PREC_F16), L.view(out, cl, cg), B, 3, 1, 1,
                  L.PAD_REFLECT, False, bias.to(DEV), L.ACT_RELU, L.view(rd), x2=L.view(td),
                  w2_packed=anylib.pack_conv_weight(w2.to(DEV), sc, precision=L.PREC_F16), precision=L.PREC_F16,
                  stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    err = float((out[:, cl:].float().cpu() - ref).abs().max())
    assert err < (2e-3 if odt == torch.float16 else 2e-4) * max(1.0, float(ref.abs().max())), err
    assert float(out[:, :cl].float().abs().max()) == 0.0


F16_BIG = [
    dict(cin=512, cout=128, k=3, stride=1, pad=1, H=64, W=64, act=1, bias=True, resid=True, scale=True, wscale=0.02),     # local conv (wreg 4 x 2)
    dict(cin=384, cout=192, k=1, stride=1, pad=0, H=64, W=64, act=1, bias=True, resid=False, scale=True, wscale=0.05),    # conv1 (persistent GEMM)
    dict(cin=384, cout=384, k=1, stride=1, pad=0, H=64, W=33, act=1, bias=True, resid=False, scale=True, wscale=0.05),    # spectral GEMM
    dict(cin=64, cout=128, k=3, stride=2, pad=1, H=128, W=128, act=1, bias=True, resid=False, scale=True, wscale=0.05),   # downsample
    dict(cin=256, cout=128, k=3, stride=2, pad=1, H=32, W=48, act=1, bias=True, resid=False, scale=True, wscale=0.03, transposed=True),
]


@pytest.mark.parametrize('case', F16_BIG, ids=lambda c: f"k{c['k']}s{c['stride']}c{c['cin']}o{c['cout']}{'T' if c.get('transposed') else ''}")
def test_conv2d_fp16_io_full_size(lib, case):
    _run_f16_case_gpu(lib, case)


@pytest.mark.parametrize('xdt,ydt', MIXED_IO, ids=['f32_to_f16', 'f16_to_f32'])
@pytest.mark.parametrize('case', [F16_BIG[0], F16_BIG[1], F16_BIG[3], F16_BIG[4]], ids=['local', 'conv1', 'down', 'up'])
def test_conv2d_fp16_mixed_io_full_size(lib, case, xdt, ydt):
    if case is F16_BIG[1] and ydt == torch.float32:
        case = dict(case, resid=True)
    _run_f16_case_gpu(lib, case, xdt, ydt)


def test_conv2d_fp16_stem_head_full_size(lib):
    """the dedicated stem / head kernels (production thresholds) at the ends of the fp16 path"""
    _run_f16_case_gpu(lib, dict(cin=4, cout=64, k=7, stride=1, pad=3, H=512, W=256, act=1, bias=True, resid=False, scale=True, wscale=0.1),
                      torch.float32, torch.float16)
    _run_f16_case_gpu(lib, dict(cin=64, cout=3, k=7, stride=1, pad=3, H=512, W=416, act=2, bias=True, resid=False, scale=False, wscale=0.03),
                      torch.float16, torch.float32)


@pytest.mark.parametrize('hw', [(64, 64), (128, 128), (32, 32), (256, 256), (24, 40)], ids=lambda s: f'{s[0]}x{s[1]}')
def test_rfft2_irfft2_fp16_io(lib, hw):
    h, w = hw
    g = torch.Generator().manual_seed(h * 7 + w)
    B, Cn = 2, 8
    x = torch.randn(B, Cn, h, w, generator=g).half()
    xd = x.to(DEV)
    spec = torch.zeros(B, 2 * Cn, h, w // 2 + 1, dtype=torch.float16, device=DEV)
    ws = torch.zeros(max(lib.fft_workspace_bytes(B, Cn, h, w), 4) // 4, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    lib.rfft2(L.view(xd), L.view(spec), B, ws, stream=st)
    ref = _spec_ref(x.float())
    assert torch.allclose(spec.float().cpu(), ref, atol=2e-3 * float(ref.abs().max()), rtol=2e-3)
    spec2 = torch.relu(torch.randn(B, 2 * Cn, h, w // 2 + 1, generator=g)).half()
    resid = torch.randn(B, Cn, h, w, generator=g).half()
    y = torch.zeros(B, Cn, h, w, dtype=torch.float16, device=DEV)
    s2d, rd = spec2.to(DEV), resid.to(DEV)
    lib.irfft2(L.view(s2d), L.view(rd), L.view(y), B, ws, stream=st)
    ref2 = resid.float() + _inv_ref(spec2.float(), h, w)
    assert torch.allclose(y.float().cpu(), ref2, atol=4e-3, rtol=2e-3)


@pytest.mark.parametrize('prec', [L.PREC_BF16X3, L.PREC_F16X3], ids=['bf16x3', 'f16x3'])
@pytest.mark.parametrize('case', WINO_CASES + [dict(cin=512, cout=128, H=64, W=64, B=8, act=1, bias=True, resid=True, scale=True),
                                               dict(cin=512, cout=128, H=128, W=128, B=2, act=1, bias=True, resid=True, scale=True),
                                               dict(cin=64, cout=128, H=32, W=256, B=1, act=0, bias=False, resid=False, scale=False),
                                               # round 6: the bottleneck planes of photo-sized inputs (1080 x 1920, 1344^2, 1000 x 1504, 4 x 720 x 1280)
                                               dict(cin=512, cout=128, H=135, W=240, B=1, act=1, bias=True, resid=True, scale=True),
                                               dict(cin=512, cout=128, H=168, W=168, B=1, act=1, bias=True, resid=True, scale=True),
                                               dict(cin=512, cout=128, H=125, W=188, B=1, act=1, bias=True, resid=False, scale=True),
                                               dict(cin=512, cout=128, H=90, W=160, B=4, act=1, bias=True, resid=True, scale=True),
                                               dict(cin=64, cout=128, H=31, W=255, B=1, act=0, bias=False, resid=False, scale=False)],
                         ids=lambda c: f"c{c['cin']}o{c['cout']}_{c['H']}x{c['W']}b{c['B']}")
def test_winograd_conv3x3(lib, case, prec):
    """lama_winograd_conv3x3_fwd (Winograd F(2x2, 3x3), wino_dev.inc) against the plain torch conv and the direct HIP kernel, incl. the
    bottleneck's local conv at BASELINE configs[1] / configs[2] shapes (8 x 512 -> 128 at 64 x 64, 2 x the same at 128 x 128)."""
    g = torch.Generator().manual_seed(5)
    B, cin, cout, H, W = case['B'], case['cin'], case['cout'], case['H'], case['W']
    xbuf = torch.randn(B, cin + 2, H, W, generator=g)
    x = xbuf[:, 1:1 + cin]
    w = torch.randn(cout, cin, 3, 3, generator=g) * (0.2 if cin <= 64 else 0.03)
    scale = torch.rand(cout, generator=g) + 0.5 if case['scale'] else None
    bias = torch.randn(cout, generator=g) if case['bias'] else None
    ref0 = _conv_ref(x, w, 1, 1, True, False, None, 0, None, scale=scale)
    resid = torch.randn(ref0.shape, generator=g) if case['resid'] else None
    ref = _conv_ref(x, w, 1, 1, True, False, bias, case['act'], resid, scale=scale)
    assert lib.winograd_supported(cout, cin, H, W, prec)
    wp = lib.pack_winograd_weight(w.to(DEV), None if scale is None else scale.to(DEV), prec)
    ws = torch.zeros(lib.winograd_workspace_bytes(B, cout, H, W) // 4, device=DEV)
    ybuf = torch.full((B, cout + 3, H, W), 7.0, device=DEV)
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    xd, rd, bd = xbuf.to(DEV), None if resid is None else resid.to(DEV), None if bias is None else bias.to(DEV)
    st = torch.cuda.current_stream().cuda_stream
    lib.winograd_conv3x3(L.view(xd, 1, cin), wp, L.view(ybuf, 2, cout), B, ws, bd, case['act'], None if rd is None else L.view(rd), precision=prec,
                         stream=st, range_flag=flag)
    torch.cuda.synchronize()
    y = ybuf[:, 2:2 + cout].cpu()
    tol = dict(atol=1.2e-3, rtol=4e-4) if prec == L.PREC_BF16X3 else dict(atol=3e-4, rtol=1e-4)
    assert torch.allclose(y, ref, **tol), float((y - ref).abs().max())
    assert float(ybuf[:, :2].min()) == 7.0 and float(ybuf[:, -1].max()) == 7.0 and int(flag) == 0
    wd = lib.pack_conv_weight(w.to(DEV), None if scale is None else scale.to(DEV), precision=prec)
    yd = torch.zeros(B, cout, H, W, device=DEV)
    lib.conv2d(L.view(xd, 1, cin), wd, L.view(yd), B, 3, 1, 1, L.PAD_REFLECT, False, bd, case['act'], None if rd is None else L.view(rd), precision=prec, stream=st)
    torch.cuda.synchronize()
    err_w, err_d = float((y - ref).abs().max()), float((yd.cpu() - ref).abs().max())
    print(f'winograd max-abs {err_w:.2e} (mean {float((y - ref).abs().mean()):.2e}); direct kernel {err_d:.2e}')
    assert err_w < 4 * max(err_d, 2e-5)
