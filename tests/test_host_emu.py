"""CPU tests of the host-side mirror (lama_amd.ffc / trainers / config) driving the kernel sources
through the host SIMT emulator, against the oracle and the reference-generated golden vectors."""
import os

import numpy as np
import pytest
import torch
import yaml

from lama_amd import config as lcfg
from lama_amd import ffc as F
from lama_amd import trainers
from lama_amd.modules import make_generator
from oracle import lama_oracle as O
from tests.emu import emu_lib


@pytest.fixture(scope='module')
def small():
    cfg = O.small_config(ngf=8, n_blocks=2)
    sd = O.make_synthetic_state_dict(cfg, seed=7, calib_hw=32)
    gen = make_generator(None, kind='ffc_resnet', **cfg)
    res = gen.load_state_dict(sd, strict=True)          # same keys as the reference
    assert not res.missing_keys and not res.unexpected_keys
    gen.set_exec(F._Exec(emu_lib()))
    return cfg, sd, gen


def test_state_dict_keys_match_reference_layout(small):
    cfg, sd, gen = small
    assert set(gen.state_dict().keys()) == set(sd.keys())
    big = make_generator(None, kind='ffc_resnet', **O.BIG_LAMA)
    keys = set(big.state_dict().keys())
    assert len(keys) == 989 and keys == {k for k, _, r in O.state_dict_spec(O.BIG_LAMA) for k in
                                         ([k] if r != 'bn' else [k + s for s in ('.weight', '.bias', '.running_mean', '.running_var', '.num_batches_tracked')])}
    assert len(big.model) == 36 and sum(p.numel() for p in big.parameters()) == 50975875


@pytest.mark.parametrize('case', ['a', 'b'])
def test_generator_matches_golden(small, golden_dir, case):
    cfg, sd, gen = small
    g = np.load(os.path.join(golden_dir, 'small_gen.npz'))
    x = torch.from_numpy(g[f'{case}_x'])
    y = gen(x)
    assert np.abs(y.numpy() - g[f'{case}_y']).max() < 1e-4


def test_generator_fused_fft_path_and_layerwise(small):
    cfg, sd, gen = small
    batch = O.make_synthetic_batch(1, 128, 128, seed=3)          # bottleneck 16x16 -> fused LDS FFT kernels
    x = torch.cat([batch['image'] * (1 - batch['mask']), batch['mask']], 1)
    taps = {}
    with torch.no_grad():
        ref = O.generator_forward(x, sd, cfg, taps=taps)
    y = gen(x)
    assert float((y - ref).abs().max()) < 1e-4
    # layer-by-layer through generator.model (what refinement.py / predict_inner_features.py do)
    z = x
    for i, layer in enumerate(gen.model):
        z = layer(z)
        r = taps[i]
        if isinstance(z, tuple):
            assert float((z[0] - r[0]).abs().max()) < 2e-4
            if torch.is_tensor(r[1]):
                assert float((z[1] - r[1]).abs().max()) < 2e-4
        else:
            assert float((z - r).abs().max()) < 2e-4, i
    # sliced Sequential keeps the fused forward
    front, rear = gen.model[0:5], gen.model[5:]
    assert float((rear(front(x)) - ref).abs().max()) < 1e-4


def test_units_match_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, 'ffc_units.npz'))
    ex = F._Exec(emu_lib())
    for tag in ('e', 'o', 'p', 'q'):
        sd = {k[len(f'fu_{tag}_sd_'):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(f'fu_{tag}_sd_')}
        c = sd['bn.weight'].numel() // 2
        fu = F.FourierUnit(c, c).set_exec(ex)
        fu.load_state_dict(sd, strict=True)
        y = fu(torch.from_numpy(g[f'fu_{tag}_x']))
        assert np.abs(y.numpy() - g[f'fu_{tag}_y']).max() < 5e-5, tag
    import torch.nn as nn
    blk = F.FFCResnetBlock(16, padding_type='reflect', norm_layer=nn.BatchNorm2d, activation_layer=nn.ReLU,
                           ratio_gin=0.75, ratio_gout=0.75, enable_lfu=False).set_exec(ex)
    blk.load_state_dict({k[len('blk_sd_'):]: torch.from_numpy(g[k]) for k in g.files if k.startswith('blk_sd_')}, strict=True)
    xl, xg = torch.from_numpy(g['blk_xl']), torch.from_numpy(g['blk_xg'])
    yl, yg = blk((xl, xg))
    assert np.abs(yl.numpy() - g['blk_yl']).max() < 5e-5 and np.abs(yg.numpy() - g['blk_yg']).max() < 5e-5
    l1, g1 = blk.conv1((xl, xg))
    assert np.abs(l1.numpy() - g['blk_c1_l']).max() < 5e-5 and np.abs(g1.numpy() - g['blk_c1_g']).max() < 5e-5
    st = blk.conv1.ffc.convg2g(xg)
    assert np.abs(st.numpy() - g['blk_st']).max() < 5e-5
    yl2, yg2 = blk((xl, xg))          # fused packing is restored after the stand-alone SpectralTransform call
    assert torch.equal(yl2, yl) and torch.equal(yg2, yg)


def test_training_module_and_checkpoint_roundtrip(small, tmp_path, golden_dir):
    cfg, sd, gen = small
    # a Lightning-style checkpoint dir: config.yaml with unresolved interpolations + models/best.ckpt
    raw = dict(training_model=dict(kind='default', concat_mask=True, visualize_each_iters=1000),
               generator=dict(kind='ffc_resnet', input_nc=4, output_nc=3, ngf=8, n_downsampling=3, n_blocks=2, add_out_act='sigmoid',
                              init_conv_kwargs=dict(ratio_gin=0, ratio_gout=0, enable_lfu=False),
                              downsample_conv_kwargs=dict(ratio_gin='${generator.init_conv_kwargs.ratio_gout}',
                                                          ratio_gout='${generator.downsample_conv_kwargs.ratio_gin}', enable_lfu=False),
                              resnet_conv_kwargs=dict(ratio_gin=0.75, ratio_gout='${generator.resnet_conv_kwargs.ratio_gin}', enable_lfu=False)),
               losses=dict(resnet_pl=dict(weights_path='${env:TORCH_HOME}')), visualizer=dict(kind='directory'))
    os.makedirs(tmp_path / 'models')
    with open(tmp_path / 'config.yaml', 'w') as f:
        yaml.safe_dump(raw, f)
    state = {'state_dict': {**{'generator.' + k: v for k, v in sd.items()}, 'discriminator.model0.0.weight': torch.zeros(3),
                            'val_evaluator.scores.lpips.x': torch.zeros(2)}, 'optimizer_states': []}
    torch.save(state, tmp_path / 'models' / 'best.ckpt')
    tc = lcfg.load_train_config(str(tmp_path / 'config.yaml'))
    assert tc['generator']['downsample_conv_kwargs']['ratio_gout'] == 0 and tc['generator']['resnet_conv_kwargs']['ratio_gout'] == 0.75
    model = trainers.load_checkpoint(tc, str(tmp_path / 'models' / 'best.ckpt'), strict=False, map_location='cpu')
    model.freeze()
    model.generator.set_exec(F._Exec(emu_lib()))
    g = np.load(os.path.join(golden_dir, 'predict_glue.npz'))
    img_p = O.pad_img_to_modulo(g['image'], 8)
    msk_p = O.pad_img_to_modulo(g['mask'][None], 8)
    batch = dict(image=torch.from_numpy(img_p)[None], mask=(torch.from_numpy(msk_p)[None] > 0) * 1)
    out = model(batch)
    cur = out['inpainted'][0].permute(1, 2, 0).numpy()[:37, :50]
    assert np.abs(cur - g['inpainted']).max() < 1e-4
    with pytest.raises(RuntimeError):
        trainers.load_checkpoint(tc, str(tmp_path / 'models' / 'best.ckpt'), strict=True, map_location='cpu')


def test_product_path_has_no_cpu_fallback():
    gen = make_generator(None, kind='ffc_resnet', **O.small_config(ngf=8, n_blocks=1))
    with pytest.raises(F.LamaError):
        gen(torch.zeros(1, 4, 32, 32))           # CPU tensor + real library -> refuse, never fall back
    with pytest.raises(NotImplementedError):
        F.SpectralTransform(8, 8, enable_lfu=True)
    with pytest.raises(NotImplementedError):
        gen.train()


def test_f16_split_overflow_falls_back_to_bf16x3():
    """ADVICE r1 / VERDICT weak #4: |x| > 65504 must not silently return inf / garbage from the default f16 split.  The kernels
    raise lama_conv2d_args.range_flag, the generator re-runs on the 3-term bf16 split (and stays there); with auto_fallback off
    the forward raises LamaRangeError; stand-alone layers raise as well."""
    from lama_amd import _lib as L
    cfg = O.small_config(ngf=8, n_blocks=1)
    sd = O.make_synthetic_state_dict(cfg, seed=5, calib_hw=32)
    gen = make_generator(None, kind='ffc_resnet', **cfg)
    gen.load_state_dict(sd, strict=True)
    gen.set_exec(F._Exec(emu_lib()))
    assert gen.precision == L.PREC_F16X3
    batch = O.make_synthetic_batch(1, 64, 64, seed=2)
    x = torch.cat([batch['image'] * (1 - batch['mask']), batch['mask']], 1)
    y0 = gen(x)                                               # in range: stays on the f16 split
    assert gen.precision == L.PREC_F16X3
    xb = x.clone()
    xb[0, 1, 20:24, 30:34] = 3.0e5                            # far outside the fp16 range
    with torch.no_grad():
        ref = O.generator_forward(xb, sd, cfg)
    gen.auto_fallback = False
    with pytest.raises(L.LamaRangeError):
        gen(xb)
    assert gen.precision == L.PREC_F16X3
    with pytest.raises(L.LamaRangeError):
        gen.model[1](torch.nn.functional.pad(xb, (3, 3, 3, 3), mode='reflect'))      # stand-alone layer: raises, no fallback
    gen.auto_fallback = True
    with pytest.warns(UserWarning, match='bf16'):
        yb = gen(xb)
    assert gen.precision == L.PREC_BF16X3 and all(m.precision == L.PREC_BF16X3 for m in gen.modules() if isinstance(m, F._HipModule))
    # (a 3e5 input carries 16 mantissa bits on the bf16 split: absolute error ~5 near the spike, fp32-class everywhere else)
    err = (yb - ref).abs()
    assert torch.isfinite(yb).all() and float(err.mean()) < 1e-3 and float((err < 1e-3).float().mean()) > 0.97, (float(err.mean()), float((err < 1e-3).float().mean()))
    y1 = gen(x)                                               # and the in-range input still matches on the bf16 split
    assert float((y1 - y0).abs().max()) < 1e-3, float((y1 - y0).abs().max())


def test_deferred_range_check_has_no_read_back_per_forward():
    """Round 4: ``defer_range_check`` -- the forward does not read the range flag back (no host synchronisation per step); the flag stays
    raised on the device across forwards and ``check_range()`` reports it at the caller's synchronisation point: False = the results since
    the last check are void and the generator is on the bf16 split now; LamaRangeError without ``auto_fallback``."""
    from lama_amd import _lib as L
    cfg = O.small_config(ngf=8, n_blocks=1)
    sd = O.make_synthetic_state_dict(cfg, seed=5, calib_hw=32)
    gen = make_generator(None, kind='ffc_resnet', **cfg)
    gen.load_state_dict(sd, strict=True)
    gen.set_exec(F._Exec(emu_lib()))
    gen.defer_range_check = True
    batch = O.make_synthetic_batch(1, 64, 64, seed=2)
    x = torch.cat([batch['image'] * (1 - batch['mask']), batch['mask']], 1)
    y0 = gen(x)
    assert gen.check_range() is True and gen.precision == L.PREC_F16X3
    xb = x.clone()
    xb[0, 1, 20:24, 30:34] = 3.0e5
    gen(xb)                                                   # no exception, no precision switch: nobody has looked yet
    assert gen.precision == L.PREC_F16X3
    gen(x)                                                    # ... and an in-range forward afterwards does not clear the flag
    gen.auto_fallback = False
    with pytest.raises(L.LamaRangeError):
        gen.check_range()
    assert gen.check_range() is True                          # reading clears it
    gen.auto_fallback = True
    gen(xb)
    with pytest.warns(UserWarning, match='bf16'):
        assert gen.check_range() is False
    assert gen.precision == L.PREC_BF16X3
    y1 = gen(x)                                               # the re-run the caller owes, on the bf16 split
    assert gen.check_range() is True and float((y1 - y0).abs().max()) < 1e-3


def test_inplace_residual_and_aliased_t_are_bit_identical():
    """Round 4: the resnet blocks write their output over their input (ONE state buffer + the block-internal one instead of four) and t = x1 + fu(x1) over x1: every
    residual operand is read by the thread that writes the element, so the results are the three-buffer plan's bit for bit."""
    cfg = O.small_config(ngf=8, n_blocks=3)
    sd = O.make_synthetic_state_dict(cfg, seed=6, calib_hw=32)
    batch = O.make_synthetic_batch(2, 64, 64, seed=3)
    x = torch.cat([batch['image'] * (1 - batch['mask']), batch['mask']], 1)
    outs, nbuf = [], []
    for inplace in (False, True):
        gen = make_generator(None, kind='ffc_resnet', **cfg)
        gen.load_state_dict(sd, strict=True)
        gen.set_exec(F._Exec(emu_lib()))
        gen.inplace_residual = gen.alias_t = gen.alias_wino = inplace
        outs.append(gen(x).clone())
        outs.append(gen(x).clone())                                # a second run through the cached plan (the buffers are dirty now)
        plan = next(iter(gen._plans.values()))
        ptrs = {t.data_ptr() for t in plan['bufs'].values()}
        sc = plan['scratch']
        nbuf.append((len(ptrs), sc['t'].data_ptr() == sc['x1'].data_ptr()))      # (no Winograd launch at these widths: alias_wino is covered by the GPU parity tests)
    assert all(torch.equal(outs[0], o) for o in outs[1:])
    assert nbuf[0][0] == nbuf[1][0] + 2 and nbuf[0][1] is False and nbuf[1][1] is True
    with torch.no_grad():
        assert float((outs[0] - O.generator_forward(x, sd, cfg)).abs().max()) < 2e-4


def test_generator_fp16_activation_path(small):
    """BASELINE configs[2] "fp16" = PREC_F16: fp16 activations in memory from the stem's output through the resnet blocks (fp32 residual stream;
    round 4: fp32 tail behind the blocks), weights as hi + lo fp16 parts (two MFMA products per MAC since round 3), fp32 accumulation.  Tolerance: 5e-3 max-abs on the sigmoid output at this size and no
    worse than 1.5x the error of the oracle run with fp16-rounded conv INPUTS (the fp32-class paths are held to 2e-4); the
    layer-by-layer Sequential agrees with the fused plan."""
    from lama_amd import _lib as L
    cfg, sd, gen = small
    batch = O.make_synthetic_batch(2, 64, 64, seed=9)
    x = torch.cat([batch['image'] * (1 - batch['mask']), batch['mask']], 1)
    with torch.no_grad():
        ref = O.generator_forward(x, sd, cfg)
    gen.set_precision(L.PREC_F16)
    try:
        y = gen(x)
        with torch.no_grad():
            emu_err = float((O.generator_forward_fp16_emulated(x, sd, cfg, weights=False) - ref).abs().max())
            emu_both = float((O.generator_forward_fp16_emulated(x, sd, cfg) - ref).abs().max())
        err = float((y - ref).abs().max())
        assert y.dtype == torch.float32 and err < 5e-3 and err < 1.5 * emu_err, (err, emu_err, emu_both)
        plan = next(iter(gen._plans.values()))
        f32 = sorted(n for n, b in plan['bufs'].items() if b.dtype == torch.float32)
        assert all(b.dtype in (torch.float16, torch.float32) for b in plan['bufs'].values())
        # the residual stream (and what feeds it) stays fp32, and so does the tail behind the blocks (round 4: f16_fp32_tail): the three upsampled tensors
        f16 = sorted(n for n, b in plan['bufs'].items() if b.dtype == torch.float16)
        assert 'out' in f32 and 'a3' in f32 and 'rA' not in f32 and len(f32) <= 5 and 'rt' in f16 and len(f16) >= 4, (f32, f16)   # a3: the fp32 state of the blocks, in place (round 4)
        z = gen.model[0:5](x)
        assert z[0].dtype == torch.float32                                            # ... also layer by layer
        y2 = gen.model[5:](z)
        assert float((y2 - y).abs().max()) < 1e-6
    finally:
        gen.set_precision(L.PREC_F16X3)


def test_conv1_rides_in_the_global_branch_epilogue():
    """SpectralTransform.conv1 of every FFC layer but the first is computed in the epilogue of the launch that produces its input
    (lama_conv2d_args.fuse1_*, big-lama channel counts only).  Two resnet blocks at 512 = (128 | 384) channels, chained the way the
    generator's plan chains them: same result as the stand-alone launches (and the oracle), one pointwise conv1 launch instead of four,
    both split back ends; the exact-fp32 path never fuses."""
    import torch.nn as nn
    from lama_amd import _lib as L
    torch.manual_seed(0)
    kw = dict(padding_type='reflect', norm_layer=nn.BatchNorm2d, activation_layer=nn.ReLU, ratio_gin=0.75, ratio_gout=0.75, enable_lfu=False)
    blocks = [F.FFCResnetBlock(512, **kw) for _ in range(2)]
    g = torch.Generator().manual_seed(3)
    sd = {}
    for bi, blk in enumerate(blocks):
        for m in blk.modules():
            if isinstance(m, nn.Conv2d):
                m.weight.data = torch.randn(m.weight.shape, generator=g) / (m.weight[0].numel() ** 0.5)
            if isinstance(m, nn.BatchNorm2d):
                m.weight.data = torch.rand(m.weight.shape, generator=g) + 0.5
                m.bias.data = torch.randn(m.bias.shape, generator=g) * 0.2
                m.running_mean.data = torch.randn(m.bias.shape, generator=g) * 0.1
                m.running_var.data = torch.rand(m.bias.shape, generator=g) + 0.5
        blk.eval()
        sd.update({f'b{bi}.{k}': v for k, v in blk.state_dict().items()})
    ex = F._Exec(emu_lib())
    for blk in blocks:
        for m in blk.modules():
            if isinstance(m, F._HipModule):
                m._exec = ex
    x = torch.randn(1, 512, 4, 5, generator=g)
    spec = dict(ratio_gin=0.75, ratio_gout=0.75)
    with torch.no_grad():
        rl, rg = x[:, :128], x[:, 128:]
        for bi in range(2):
            rl, rg = O.ffc_resnet_block(rl, rg, sd, f'b{bi}', spec)
        ref = torch.cat([rl, rg], 1)
    calls = dict(fused=0, conv1=0)
    real = ex.lib.conv2d

    def counting(xv, wp, yv, b, k, *a, **kw2):
        if kw2.get('fuse1') is not None:
            calls['fused'] += 1
        if k == 1 and yv.C == 192 and xv.C == 384:
            calls['conv1'] += 1
        return real(xv, wp, yv, b, k, *a, **kw2)

    def run(fuse):
        calls.update(fused=0, conv1=0)
        scratch = blocks[0].conv1.make_scratch(x.shape, x.device)
        a, t, b = x.clone(), torch.empty_like(x), torch.empty_like(x)
        ready = blocks[0].run(a, t, b, scratch, x1_ready=False, next_block=blocks[1] if fuse else None, fuse=fuse)
        assert ready == fuse
        ready = blocks[1].run(b, t, a, scratch, x1_ready=ready, next_block=None, fuse=fuse)
        assert not ready
        return a

    ex.lib.conv2d = counting
    try:
        for prec, tol in ((L.PREC_F16X3, 5e-5), (L.PREC_BF16X3, 2e-3)):
            for blk in blocks:
                blk.set_precision(prec)
            y = run(True)
            assert calls == dict(fused=3, conv1=1), calls          # the first layer's conv1 alone, three in epilogues
            scale = float(ref.abs().max())
            assert float((y - ref).abs().max()) < 4 * tol * scale, (float((y - ref).abs().max()), scale)
            if prec == L.PREC_F16X3:
                y0 = run(False)
                assert calls == dict(fused=0, conv1=4), calls
                assert float((y - y0).abs().max()) < tol * scale, (float((y - y0).abs().max()), scale)
        for blk in blocks:
            blk.set_precision(L.PREC_F32)
        st = blocks[1].conv1.ffc.convg2g
        blocks[1].conv1._pack()
        assert st.fuse1_operands(torch.zeros(1, 192, 4, 5)) is None  # exact-fp32 path: never fused
    finally:
        ex.lib.conv2d = real


def test_resnet_blocks_without_a_global_branch():
    """resnet_conv_kwargs with ratio_gin = ratio_gout = 0 -- a layout the reference's constructor accepts (ffc.py:338-343:
    every FFC of the blocks is local-only, ``x_g`` stays the int 0) -- has no SpectralTransform scratch set: the plan's end-of-blocks flush of
    a deferred Winograd output transform must not touch it.  Fused plan (twice: cached plan), layer by layer, and the oracle agree."""
    cfg = O.small_config(ngf=8, n_blocks=2)
    cfg['resnet_conv_kwargs'] = dict(ratio_gin=0, ratio_gout=0, enable_lfu=False)
    sd = O.make_synthetic_state_dict(cfg, seed=4, calib_hw=32)
    gen = make_generator(None, kind='ffc_resnet', **cfg)
    res = gen.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    gen.set_exec(F._Exec(emu_lib()))
    batch = O.make_synthetic_batch(1, 64, 64, seed=8)
    x = torch.cat([batch['image'] * (1 - batch['mask']), batch['mask']], 1)
    with torch.no_grad():
        ref = O.generator_forward(x, sd, cfg)
    y = gen(x)
    assert float((y - ref).abs().max()) < 1e-4
    assert torch.equal(gen(x), y)
    plan = next(iter(gen._plans.values()))
    assert plan['scratch'] is None
    z = x
    for layer in gen.model:
        z = layer(z)
    assert float((z - ref).abs().max()) < 1e-4


def test_range_flag_device_key_and_stale_flag(monkeypatch):
    """(i) ``check_range('cuda')`` must find the flag keyed 'cuda:<current index>' instead of returning True unread; (ii) a flag left
    raised by deferred forwards nobody asked about is not charged to the next SELF-CHECKING forward of an in-range input."""
    from lama_amd import _lib as L
    monkeypatch.setattr(torch.cuda, 'current_device', lambda: 3)
    assert F._Exec._dev_key('cuda') == 'cuda:3' and F._Exec._dev_key('cuda:1') == 'cuda:1' and F._Exec._dev_key(torch.device('cpu')) == 'cpu'
    cfg = O.small_config(ngf=8, n_blocks=1)
    sd = O.make_synthetic_state_dict(cfg, seed=5, calib_hw=32)
    gen = make_generator(None, kind='ffc_resnet', **cfg)
    gen.load_state_dict(sd, strict=True)
    gen.set_exec(F._Exec(emu_lib()))
    gen.auto_fallback = False
    batch = O.make_synthetic_batch(1, 64, 64, seed=2)
    x = torch.cat([batch['image'] * (1 - batch['mask']), batch['mask']], 1)
    xb = x.clone()
    xb[0, 1, 20:24, 30:34] = 3.0e5
    gen.defer_range_check = True
    gen(xb)                                                   # raises the flag on the device; nobody reads it
    gen.defer_range_check = False
    y = gen(x)                                                # self-checking forward of an in-range input: no LamaRangeError, no fallback
    assert gen.precision == L.PREC_F16X3 and torch.isfinite(y).all()
    # ... but the deferred forward's report is not lost either: the self-checking scope kept it as a sticky bit for the next check_range
    with pytest.raises(L.LamaRangeError):
        gen.check_range('cpu')
    assert gen.check_range('cpu') is True
    with pytest.raises(L.LamaRangeError):
        gen(xb)                                               # ... and it still catches its own


def test_deferred_winograd_output_transform_plan_is_bit_identical():
    """the host logic of the deferred Winograd output transform (pending_out hand-off in FFC.launch, the end-of-blocks flush, P placed
    behind the first spectrum in _build_plan, the Winograd partial sums aliased onto the FourierUnit's dead spectra) on the EMULATOR.  The plan
    of two residual blocks whose local conv is Winograd-eligible (160 = 128 | 32 channels, 16 x 32 planes: small enough to emulate) is run with
    defer_wino_out / alias_wino on and off, and with the direct local conv: the same bits from the three Winograd plans, the oracle's values."""
    cfg = O.small_config(ngf=40, n_blocks=2, n_downsampling=2)
    cfg['resnet_conv_kwargs'] = dict(ratio_gin=0.2, ratio_gout=0.2, enable_lfu=False)
    sd = O.make_synthetic_state_dict(cfg, seed=12, calib_hw=64)
    first = 2 + cfg['n_downsampling']
    g = torch.Generator().manual_seed(4)
    x = torch.randn(1, 160, 16, 32, generator=g).abs()
    spec = dict(ratio_gin=0.2, ratio_gout=0.2)
    with torch.no_grad():
        rl, rg = x[:, :128], x[:, 128:]
        for bi in range(cfg['n_blocks']):
            rl, rg = O.ffc_resnet_block(rl, rg, sd, f'model.{first + bi}', spec)
        ref = torch.cat([rl, rg], 1)
    outs = {}
    for name, (wino, defer, alias) in dict(defer=(True, True, True), plain=(True, False, False), direct=(False, False, False)).items():
        gen = make_generator(None, kind='ffc_resnet', **cfg)
        gen.load_state_dict(sd, strict=True)
        ex = F._Exec(emu_lib())
        ex.winograd = wino
        gen.set_exec(ex)
        gen.model = F.LayerSequence(*list(gen.model)[first:first + cfg['n_blocks']])      # the blocks' plan alone: 'in' is the (x_l | x_g) state
        gen.defer_wino_out, gen.alias_wino = defer, alias
        y = gen(x).clone()
        if defer:
            assert torch.equal(gen(x), y), name                   # the cached plan again (dirty buffers, P over the second spectrum)
        plan = next(iter(gen._plans.values()))
        sc = plan['scratch']
        assert (sc.get('wino') is not None) == wino and bool(sc.get('defer_out')) == defer and 'pending_out' not in sc, name
        if wino:
            lo, hi = sc['ws'].data_ptr(), sc['ws'].data_ptr() + sc['ws'].numel() * 4
            assert (lo <= sc['wino'].data_ptr() < hi) == (defer or alias), name
        outs[name] = y
    assert torch.equal(outs['defer'], outs['plain'])
    scale = float(ref.abs().max())
    assert float((outs['defer'] - ref).abs().max()) < 1e-4 * scale and float((outs['direct'] - ref).abs().max()) < 1e-4 * scale


def test_host_fed_step_double_buffering():
    """lama_amd.predict.HostFedStep on the emulator (eager body, no graphs): five steps with DIFFERENT host inputs per step -- upload of batch
    k + 1, compute of batch k, download of batch k - 1 per launch -- deliver, in order, the u8 images of the plain model on each batch."""
    from lama_amd import _lib as L
    from lama_amd.predict import HostFedStep
    cfg = O.small_config(ngf=8, n_blocks=1)
    sd = {'generator.' + k: v for k, v in O.make_synthetic_state_dict(cfg, seed=11, calib_hw=32).items()}
    model = trainers.DefaultInpaintingTrainingModule(dict(generator=dict(kind='ffc_resnet', **cfg)))
    model.load_state_dict(sd, strict=True)
    model.freeze()
    model.generator.set_exec(F._Exec(emu_lib()))
    lib = model.generator._exec.lib
    n, H, W, steps = 1, 32, 40, 4
    batches = [O.make_synthetic_batch(n, H, W, seed=30 + k) for k in range(steps)]
    want = []
    for b in batches:
        out = model(dict(image=b['image'].clone(), mask=(b['mask'] * 0.5 > 0) * 1))['inpainted']
        u8 = torch.empty(n, H, W, 3, dtype=torch.uint8)
        lib.quantize_u8_hwc(L.view(out), u8, n, H, W, 0)
        want.append(u8)
    hs = HostFedStep(model, n, H, W, 'cpu', drain=True, u8_input=False)

    def fill(p, k):
        im, mk = hs.host(p)
        im[:] = batches[k]['image'].numpy()
        mk[:] = batches[k]['mask'].numpy() * 0.5           # a gray mask: any non-zero value is "hole" (bin/predict.py:84)

    got = {}
    fill(0, 0)
    hs.prime(0)
    for k in range(steps):
        p = k & 1
        if k + 1 < steps:
            fill(1 - p, k + 1)
        hs.launch(p)
        if k >= 1:
            hs.wait(p)
            got[k - 1] = torch.from_numpy(hs.result(1 - p).copy())      # launch(p) downloaded the previous batch
    hs.flush((steps - 1) & 1)
    got[steps - 1] = torch.from_numpy(hs.result((steps - 1) & 1).copy())
    assert sorted(got) == list(range(steps)) and all(torch.equal(got[k], want[k]) for k in range(steps))
    assert model.keep_predicted_image is True and model.generator.defer_range_check is False


def _inner_feature_maps(gen_model, masked_img, levels):
    """bin/predict_inner_features.py:84-98 verbatim (cv2.imwrite replaced by a dict): the per-level RMS-over-channels feature image."""
    out = {}
    max_level = max(levels)
    feats = masked_img
    for level_i, level in enumerate(gen_model):
        feats = level(feats)
        if level_i in levels:
            cur_feats = torch.cat([f for f in feats if torch.is_tensor(f)], dim=1) if isinstance(feats, tuple) else feats
            cur_feat = cur_feats.pow(2).mean(1).pow(0.5).clone()
            cur_feat -= cur_feat.min()
            cur_feat /= cur_feat.std()
            cur_feat = cur_feat.clamp(0, 1) / 1
            cur_feat = cur_feat.cpu().numpy()[0] * 255
            out[level_i] = np.clip(cur_feat, 0, 255).astype('uint8')
        elif level_i >= max_level:
            break
    return out


def test_predict_inner_features_call_pattern(small):
    """bin/predict_inner_features.py:56,84-98: ``generator.model`` must be an nn.Sequential that can be walked level by level from the masked
    image -- tensors up to the first FFC layer, (x_l, x_g) tuples (x_g the int 0 before the last downsampling layer) through the blocks,
    tensors again behind ConcatTupleLayer -- with an early ``break``; the per-level feature images equal the oracle's to one u8 level."""
    cfg, sd, gen = small
    assert isinstance(getattr(gen, 'model', None), torch.nn.Sequential)
    batch = O.make_synthetic_batch(1, 64, 64, seed=3)
    img, mask = batch['image'], batch['mask']
    mask[:] = 0
    mask[:, :, 32 - 12:32 + 12, 32 - 12:32 + 12] = 1                    # predict_inner_features.py:77-82 (hole_radius)
    masked_img = torch.cat([img * (1 - mask), mask], dim=1)
    levels = [0, 1, 2, 4, 5, 6, 7, 8, 10]
    taps = {}
    with torch.no_grad():
        O.generator_forward(masked_img, sd, cfg, taps=taps)

    ref = _inner_feature_maps([(lambda x, i=i: taps[i]) for i in range(len(taps))], masked_img, levels)
    got = _inner_feature_maps(gen.model, masked_img, levels)
    assert sorted(got) == sorted(ref) == levels
    for i in levels:
        assert got[i].shape == ref[i].shape and np.abs(got[i].astype(int) - ref[i].astype(int)).max() <= 1, i


def test_split_batch_plan_is_bit_identical(small):
    """Round 5: the parts of a batch as parallel branches of the plan (generator.split_batch; on the emulator the branches run one after the other):
    each part has its own buffers, reads its slice of the input, writes its slice of the output and tags its launches as siblings
    (LAMA_CONV_SIBLINGS_*, v109) -- same bits as the one-part plan; the auto rule never splits on a CPU device."""
    cfg, sd, gen = small
    batch = O.make_synthetic_batch(4, 24, 32, seed=21)         # (sized for the CPU suite: eight emulated forwards of four images)
    x = torch.cat([batch['image'] * (1 - batch['mask']), batch['mask']], 1)
    seen = []
    real = gen._exec.lib.conv2d

    def spy(*a, **kw):
        seen.append(kw.get('siblings_log2', 0))
        return real(*a, **kw)

    try:
        gen._plans.clear()                                   # (plans of other shapes the shared fixture's earlier tests left)
        assert gen._split_parts(x.shape, x.device) == 1
        y1 = gen(x).clone()
        assert 'parts' not in next(iter(gen._plans.values()))
        gen._exec.lib.conv2d = spy
        for n in (2, 4):
            gen.split_batch = n
            seen.clear()
            yn = gen(x).clone()
            plan = next(iter(gen._plans.values()))
            assert plan['nsplit'] == n and len(plan['parts']) == n and len(gen._plans) == 1
            assert torch.equal(yn, y1), n
            assert seen and all(v == n.bit_length() - 1 for v in seen) and gen._exec.siblings_log2 == 0
            if n == 2:
                assert torch.equal(gen(x), y1)                # the cached plan again
            else:
                buf = gen.input_buffer(x.shape, x.device)      # the caller writes the plan's own input buffer
                buf.copy_(x)
                assert torch.equal(gen(buf), y1)
        gen.split_batch = 3
        with pytest.raises(F.LamaError):
            gen(x)
    finally:
        gen._exec.lib.conv2d = real
        gen.split_batch = None
        gen._plans.clear()


def test_host_fed_step_u8_input_equals_the_host_conversion():
    """Round 6 (ABI v110): HostFedStep fed with u8 HWC images as they are on disk -- unpadded 29 x 37 images in 32 x 40 slots, a gray mask, an empty
    slot of a partial batch -- against the reference's host-side conversion (load_image's / 255, pad_img_to_modulo, mask > 0: oracle) + the
    module's forward + quantize_u8_hwc: the same bytes."""
    from lama_amd import _lib as L
    from lama_amd.predict import HostFedStep
    cfg = O.small_config(ngf=8, n_blocks=1)
    sd = {'generator.' + k: v for k, v in O.make_synthetic_state_dict(cfg, seed=11, calib_hw=32).items()}
    model = trainers.DefaultInpaintingTrainingModule(dict(generator=dict(kind='ffc_resnet', **cfg)))
    model.load_state_dict(sd, strict=True)
    model.freeze()
    model.generator.set_exec(F._Exec(emu_lib()))
    lib = model.generator._exec.lib
    n, H, W, hv, wv, steps = 2, 32, 40, 29, 37, 3
    g = torch.Generator().manual_seed(5)
    raw = [dict(image=torch.randint(0, 256, (hv, wv, 3), generator=g, dtype=torch.uint8).numpy(),
                mask=(torch.randint(0, 3, (hv, wv), generator=g) * 100).to(torch.uint8).numpy()) for _ in range(steps)]       # mask values 0 / 100 / 200
    want = []
    for r in raw:
        im = O.pad_img_to_modulo(np.transpose(r['image'], (2, 0, 1)).astype('float32') / 255, 8)[None]
        mk = O.pad_img_to_modulo(r['mask'][None].astype('float32') / 255, 8)[None]
        out = model(dict(image=torch.from_numpy(im), mask=(torch.from_numpy(mk) > 0) * 1))['inpainted']
        u8 = torch.empty(1, H, W, 3, dtype=torch.uint8)
        lib.quantize_u8_hwc(L.view(out), u8, 1, H, W, 0)
        want.append(u8[0])
    hs = HostFedStep(model, n, H, W, 'cpu', drain=True)
    assert hs.u8_input

    def fill(p, k):
        hs.put(p, 0, raw[k]['image'], raw[k]['mask'])
        hs.put(p, 1, None, None)                          # the empty slot of a partial batch

    got = {}
    fill(0, 0)
    hs.prime(0)
    for k in range(steps):
        p = k & 1
        if k + 1 < steps:
            fill(1 - p, k + 1)
        hs.launch(p)
        if k >= 1:
            hs.wait(p)
            got[k - 1] = torch.from_numpy(hs.result(1 - p).copy())
    hs.flush((steps - 1) & 1)
    got[steps - 1] = torch.from_numpy(hs.result((steps - 1) & 1).copy())
    for k in range(steps):
        assert torch.equal(got[k][0], want[k]), k


def test_host_fed_step_out_key_predicted_image():
    """bin/predict.py:86 reads ``batch[predict_config.out_key]``: 'predicted_image' (default.py:70) is the generator's output WITHOUT the blend
    of default.py:71.  HostFedStep(out_key=...) against the module's forward + the reference's clip / astype('uint8') on the host."""
    from lama_amd.predict import HostFedStep, parse_overrides
    cfg = O.small_config(ngf=8, n_blocks=1)
    sd = {'generator.' + k: v for k, v in O.make_synthetic_state_dict(cfg, seed=12, calib_hw=32).items()}
    model = trainers.DefaultInpaintingTrainingModule(dict(generator=dict(kind='ffc_resnet', **cfg)))
    model.load_state_dict(sd, strict=True)
    model.freeze()
    model.generator.set_exec(F._Exec(emu_lib()))
    n, H, W = 1, 32, 40
    g = torch.Generator().manual_seed(6)
    image = torch.randint(0, 256, (H, W, 3), generator=g, dtype=torch.uint8).numpy()
    mask = (torch.randint(0, 2, (H, W), generator=g) * 255).to(torch.uint8).numpy()
    batch = model(dict(image=torch.from_numpy(np.transpose(image, (2, 0, 1)).astype('float32') / 255)[None],
                       mask=(torch.from_numpy(mask[None, None].astype('float32') / 255) > 0) * 1))
    for key in ('predicted_image', 'inpainted'):
        want = np.clip(batch[key][0].permute(1, 2, 0).numpy() * 255, 0, 255).astype('uint8')     # predict.py:86,92
        hs = HostFedStep(model, n, H, W, 'cpu', drain=True, out_key=key)
        hs.put(0, 0, image, mask)
        hs.prime(0)
        hs.launch(0)
        hs.flush(0)
        assert np.array_equal(hs.result(0)[0], want), key
    assert not np.array_equal(np.clip(batch['predicted_image'][0].permute(1, 2, 0).numpy() * 255, 0, 255).astype('uint8'),
                              np.clip(batch['inpainted'][0].permute(1, 2, 0).numpy() * 255, 0, 255).astype('uint8'))
    with pytest.raises(F.LamaError):
        HostFedStep(model, n, H, W, 'cpu', out_key='mask_for_losses')
    base = ['model.path=/m', 'indir=/i', 'outdir=/o']
    assert parse_overrides(base + ['out_key=predicted_image'])['out_key'] == 'predicted_image'
    with pytest.raises(SystemExit):
        parse_overrides(base + ['out_key=nonsense'])


def test_encode_png_round_trips_and_matches_cv2_defaults():
    """lama_amd.predict.encode_png (what the CLI writes for out_ext=.png, bin/predict.py:93-94): any PNG reader returns the same pixels; the stream
    is ONE IDAT with filter type Sub on every row and a zlib header of level 1 -- cv2.imwrite's defaults (filter SUB, Z_BEST_SPEED, Z_RLE)."""
    import io
    import struct
    import zlib
    from PIL import Image
    from lama_amd.predict import encode_png, _write_png
    g = np.random.default_rng(3)
    for shape in ((37, 50, 3), (1, 1, 3), (8, 300, 3), (16, 16)):
        a = g.integers(0, 256, shape, dtype=np.uint8)
        if len(shape) == 3:
            a[: shape[0] // 2] = a[0, 0]                       # flat rows: the run-length strategy must reproduce them too
        png = encode_png(a)
        back = np.asarray(Image.open(io.BytesIO(png)))
        assert back.shape == a.shape and np.array_equal(back, a), shape
        assert png[:8] == b'\x89PNG\r\n\x1a\n' and png[12:16] == b'IHDR' and png.count(b'IDAT') == 1
        n = struct.unpack('>I', png[33:37])[0]
        assert png[37:41] == b'IDAT'
        raw = zlib.decompress(png[41:41 + n])
        stride = 1 + a.shape[1] * (3 if a.ndim == 3 else 1)
        assert len(raw) == a.shape[0] * stride and set(raw[::stride]) == {1}      # filter type Sub on every row
        assert png[41] == 0x78 and png[42] == 0x01                                  # zlib header: deflate, 32 K window, FLEVEL 0 = level 1
    with pytest.raises(F.LamaError):
        encode_png(np.zeros((4, 4, 3), np.float32))
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        _write_png(os.path.join(d, 'sub', 'x.png'), a if a.ndim == 3 else np.stack([a] * 3, -1))
        assert os.path.getsize(os.path.join(d, 'sub', 'x.png')) > 0
