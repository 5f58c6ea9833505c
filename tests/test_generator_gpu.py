"""GPU parity of the whole generator / predict arithmetic against the CPU oracle and the golden
vectors produced by the reference's own classes.  Tolerance: 1e-3 max-abs fp32 on `inpainted`
(BASELINE.json north_star); the exact-fp32 MFMA path is held to 2e-4."""
import os

import numpy as np
import pytest
import torch

from lama_amd import trainers
from lama_amd.modules import make_generator
from oracle import lama_oracle as O

from lama_amd import _lib as L

pytestmark = pytest.mark.gpu
# north_star bar: 1e-3 max-abs fp32.  Held here: 2e-4 for the exact-fp32 MFMA path and the default 3-term fp16 split (22
# mantissa bits), 5e-4 for the 3-term bf16 split (16 mantissa bits; measured 1e-4 .. 3.1e-4 on the big-lama fixture).
TOLS = {L.PREC_F32: 2e-4, L.PREC_F16X3: 2e-4, L.PREC_BF16X3: 5e-4}


@pytest.fixture(scope='module', params=[L.PREC_F32, L.PREC_F16X3, L.PREC_BF16X3], ids=['f32', 'f16x3', 'bf16x3'])
def prec(request):
    return request.param


@pytest.fixture(scope='module')
def small_gen(prec):
    cfg = O.small_config(ngf=8, n_blocks=2)
    sd = O.make_synthetic_state_dict(cfg, seed=7, calib_hw=32)
    gen = make_generator(None, kind='ffc_resnet', **cfg)
    gen.load_state_dict(sd, strict=True)
    return cfg, sd, gen.cuda().set_precision(prec), TOLS[prec]


_BIG_SD = {}
_ORACLE = {}


def _oracle_big(batch_n, H, W, seed):
    """big-lama oracle output for a seeded synthetic batch, computed ONCE per (shape, seed) and shared by the three precisions
    (the CPU pass is the slow part of these tests: a few seconds per 512 x 512 image on the GPU box's host)."""
    key = (batch_n, H, W, seed)
    if key not in _ORACLE:
        torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))      # 128+ threads oversubscribe oneDNN on the GPU box
        batch = O.make_synthetic_batch(batch_n, H, W, seed=seed)
        x = torch.cat([batch['image'] * (1 - batch['mask']), batch['mask']], 1)
        with torch.no_grad():
            ref = torch.cat([O.generator_forward(x[i:i + 1], _BIG_SD['sd'], O.BIG_LAMA) for i in range(batch_n)], 0)
        _ORACLE[key] = (x, ref)
    return _ORACLE[key]


@pytest.fixture(scope='module')
def big(prec):
    cfg = O.BIG_LAMA
    if 'sd' not in _BIG_SD:
        _BIG_SD['sd'] = O.make_synthetic_state_dict(cfg, seed=0, calib_hw=64)
    sd = _BIG_SD['sd']
    gen = make_generator(None, kind='ffc_resnet', **cfg)
    gen.load_state_dict(sd, strict=True)
    return cfg, sd, gen.cuda().set_precision(prec), TOLS[prec]


@pytest.mark.parametrize('case', ['a', 'b', 'c'])
def test_small_generator_golden(small_gen, golden_dir, case):
    cfg, sd, gen, TOL = small_gen
    g = np.load(os.path.join(golden_dir, 'small_gen.npz'))
    y = gen(torch.from_numpy(g[f'{case}_x']).cuda())
    assert np.abs(y.cpu().numpy() - g[f'{case}_y']).max() < TOL


def test_small_generator_layerwise_and_sliced(small_gen):
    cfg, sd, gen, TOL = small_gen
    batch = O.make_synthetic_batch(2, 128, 128, seed=3)
    x = torch.cat([batch['image'] * (1 - batch['mask']), batch['mask']], 1)
    taps = {}
    with torch.no_grad():
        ref = O.generator_forward(x, sd, cfg, taps=taps)
    xd = x.cuda()
    assert float((gen(xd).cpu() - ref).abs().max()) < TOL
    z = xd
    for i, layer in enumerate(gen.model):
        z = layer(z)
        zz = z[0] if isinstance(z, tuple) else z
        rr = taps[i][0] if isinstance(taps[i], tuple) else taps[i]
        assert float((zz.cpu() - rr).abs().max()) < 1e-3, i
    assert float((gen.model[5:](gen.model[0:5](xd)).cpu() - ref).abs().max()) < TOL


def test_biglama_256_golden_and_oracle(big, golden_dir):
    """big-lama shape, 1x256x256: against the reference-generated samples and the full oracle output."""
    cfg, sd, gen, TOL = big
    g = np.load(os.path.join(golden_dir, 'biglama_256.npz'))
    batch = O.make_synthetic_batch(1, 256, 256, seed=1234)
    x = torch.cat([batch['image'] * (1 - batch['mask']), batch['mask']], 1)
    y = gen(x.cuda()).cpu()
    assert np.abs(y[:, :, ::8, ::8].numpy() - g['y_sample']).max() < TOL
    with torch.no_grad():
        ref = O.generator_forward(x, sd, cfg)
    assert float((y - ref).abs().max()) < TOL


def test_biglama_c1_config_and_graph(big):
    """BASELINE configs[0] shape (4x256x256) through the training-module forward (mask compose + blend),
    eager launches and hipGraph replay must agree with the oracle and with each other."""
    cfg, sd, gen, TOL = big
    batch = O.make_synthetic_batch(4, 256, 256, seed=77)
    sdg = {'generator.' + k: v for k, v in sd.items()}
    with torch.no_grad():
        ref = O.training_module_forward(dict(image=batch['image'].clone(), mask=batch['mask'].clone()), sdg, cfg)
    model = trainers.DefaultInpaintingTrainingModule(dict(generator=dict(kind='ffc_resnet', **cfg)))
    model.load_state_dict(sdg, strict=True)
    model.freeze().cuda()
    model.generator.set_precision(gen.precision)
    out = model(dict(image=batch['image'].cuda(), mask=batch['mask'].cuda()))
    assert float((out['inpainted'].cpu() - ref['inpainted']).abs().max()) < TOL
    model.generator.use_graph = True
    out2 = model(dict(image=batch['image'].cuda(), mask=batch['mask'].cuda()))
    out3 = model(dict(image=batch['image'].cuda(), mask=batch['mask'].cuda()))
    assert torch.equal(out2['inpainted'], out3['inpainted'])
    assert float((out2['inpainted'].cpu() - ref['inpainted']).abs().max()) < TOL


def test_biglama_512_batch8_all_images(big):
    """BASELINE configs[1] (8x512x512): EVERY image of the batch against the oracle, plus the size-independent properties:
    batch independence (each image equals its batch-1 run), determinism, range."""
    cfg, sd, gen, TOL = big
    x, ref = _oracle_big(8, 512, 512, 99)
    xd = x.cuda()
    y = gen(xd)
    y1 = gen(xd[3:4].contiguous())
    # batch independence up to the summation order: at batch 1 the launches are smaller and other kernels are selected (conv1 as a launch
    # of its own instead of riding in the global-branch epilogue); the 16-bit-mantissa products of bf16x3 make that visible
    assert float((y[3:4] - y1).abs().max()) < {L.PREC_F32: 1e-5, L.PREC_F16X3: 3e-5, L.PREC_BF16X3: 3e-4}[gen.precision]
    assert torch.equal(gen(xd), y)
    assert bool(torch.isfinite(y).all()) and float(y.min()) >= 0 and float(y.max()) <= 1
    err = (y.cpu() - ref).abs().amax(dim=(1, 2, 3))
    assert float(err.max()) < TOL, err.tolist()
    gen.use_graph = True                      # the captured-graph replay of the same plan
    try:
        yg = gen(xd)
        assert torch.equal(gen(xd), yg) and float((yg.cpu() - ref).abs().max()) < TOL
    finally:
        gen.use_graph = False
    # conv1 of the next layer in the global-branch epilogue (optional since 4.12): same values up to summation order
    gen.fuse_conv1 = True
    gen._plans.clear()
    try:
        yf = gen(xd)
        assert torch.equal(gen(xd), yf) and float((yf.cpu() - ref).abs().max()) < TOL
        assert float((yf - y).abs().max()) < {L.PREC_F32: 1e-5, L.PREC_F16X3: 3e-5, L.PREC_BF16X3: 3e-4}[gen.precision]
    finally:
        gen.fuse_conv1 = False
        gen._plans.clear()


def test_biglama_in_place_plan_is_bit_identical(big):
    """Round 4: the resnet blocks in place, t over x1, the Winograd partial sums in the FourierUnit's spectra (348 -> 211 MB per residual layer,
    DESIGN.md 3) -- every overwritten operand is read by the thread that writes it, so the four-buffer plan gives the SAME bits; 4 x 512^2
    (128 pixel tiles: the fused conv1 / Winograd one-stream plan of the split precisions), eager and graph replay, twice (dirty buffers)."""
    cfg, sd, gen, TOL = big
    batch = O.make_synthetic_batch(4, 512, 512, seed=41)
    xd = torch.cat([batch['image'] * (1 - batch['mask']), batch['mask']], 1).cuda()
    outs = []
    try:
        for flag in (False, True):
            gen.inplace_residual = gen.alias_t = gen.alias_wino = gen.defer_wino_out = flag
            gen._plans.clear()
            outs.append(gen(xd).clone())
            outs.append(gen(xd).clone())
            plan = next(iter(gen._plans.values()))
            sc = plan['scratch']
            if flag:
                assert sc['t'].data_ptr() == sc['x1'].data_ptr()
                if sc.get('wino') is not None and plan['side'] is None:      # P inside the FourierUnit's workspace: behind the first spectrum
                    assert sc['ws'].data_ptr() <= sc['wino'].data_ptr() < sc['ws'].data_ptr() + 4 * sc['ws'].numel() and sc.get('defer_out')
            gen.use_graph = True
            try:
                outs.append(gen(xd).clone())
            finally:
                gen.use_graph = False
    finally:
        gen.inplace_residual = gen.alias_t = gen.alias_wino = gen.defer_wino_out = True
        gen._plans.clear()
    assert all(torch.equal(outs[0], o) for o in outs[1:])


def test_training_module_properties_at_configs1_size():
    """DefaultInpaintingTrainingModule.forward at 8x512x512 (BASELINE configs[1]) through properties that need no CPU pass: outside the
    hole the inpainted image IS the input (bitwise: inpainted = m pred + (1 - m) img, trainers/default.py:71), inside it is the
    prediction; the generator never sees the pixels under the hole (changing them changes nothing); the u8 HWC result is the
    clipped x255 truncation of the float result (bin/predict.py:86-92)."""
    from lama_amd import trainers
    cfg = O.BIG_LAMA
    if 'sd' not in _BIG_SD:
        _BIG_SD['sd'] = O.make_synthetic_state_dict(cfg, seed=0, calib_hw=64)
    model = trainers.DefaultInpaintingTrainingModule(dict(generator=dict(kind='ffc_resnet', **cfg)))
    model.generator.load_state_dict(_BIG_SD['sd'], strict=True)
    model.freeze()
    model.cuda()
    batch = O.make_synthetic_batch(8, 512, 512, seed=4242)
    img, mask = batch['image'].cuda(), batch['mask'].cuda()
    out = model(dict(image=img, mask=mask))
    inp, pred = out['inpainted'], out['predicted_image']
    keep = (mask == 0).expand_as(img)
    assert torch.equal(inp[keep], img[keep]) and torch.equal(inp[~keep], pred[~keep])
    img2 = torch.where(keep, img, torch.rand_like(img))                  # other pixels under the hole
    out2 = model(dict(image=img2, mask=mask))
    assert torch.equal(out2['predicted_image'], pred)
    u8 = torch.empty(8, 512, 512, 3, dtype=torch.uint8, device='cuda')
    lib = model.generator._exec.lib
    lib.quantize_u8_hwc(L.view(inp), u8, 8, 512, 512, torch.cuda.current_stream().cuda_stream)
    ref_u8 = torch.clamp(inp.permute(0, 2, 3, 1) * 255, 0, 255).to(torch.uint8)
    assert torch.equal(u8, ref_u8)
    model.generator._plans.clear()


@pytest.mark.parametrize('res', [1024, 2048], ids=['1024sq_planes128', '2048sq_planes256'])
def test_biglama_high_res_square(big, res):
    """BASELINE configs[2] / configs[4] resolutions at batch 1 against the full oracle: 1024^2 -> 128 x 128 bottleneck planes
    (one-buffer LDS FFT kernels rfft2_ipn_kernel<128>), 2048^2 -> 256 x 256 planes (two-pass LDS FFT through the workspace)."""
    cfg, sd, gen, TOL = big
    x, ref = _oracle_big(1, res, res, 1000 + res)
    y = gen(x.cuda()).cpu()
    gen._plans.clear()                        # 2.2 GB (1024^2) / 8.6 GB (2048^2) of activation buffers
    # K-long sums and 256-point FFTs: the worst pixel of 4 M sits a little higher than at 512^2 (measured 2.0e-4 at 2048^2 on
    # the f16 split); held to 1.5x the 512^2 tolerance, still 3x below the 1e-3 bar
    err = float((y - ref).abs().max())
    assert err < 1.5 * TOL, err


def test_biglama_two_1024sq_spectral_gemm_one_wave(big):
    """2 x 1024^2: the spectral 1x1 of every FourierUnit is 2 x 128 x 65 points = 260 super-tiles of 64 -- the second production shape of
    gemm1x1_wk_kernel (all of K in one wave, gemm_wk_dev.inc: one round on 256 CUs + 4 left-over super-tiles = 96 units on the fifth
    waves; 8 x 512^2 has 8 = 192), here inside the whole generator against the oracle."""
    cfg, sd, gen, TOL = big
    x, ref = _oracle_big(2, 1024, 1024, 7071)
    y = gen(x.cuda()).cpu()
    gen._plans.clear()
    err = float((y - ref).abs().max())
    assert err < 1.5 * TOL, err


@pytest.mark.parametrize('shape', [(4, 1024), (2, 512), (1, 256)], ids=['c3_4x1024', '2x512', '1x256'])
def test_biglama_fp16_activation_path(shape):
    """BASELINE configs[2] (big-lama 1024x1024 batch=4 fp16): PREC_F16 = fp16 activations in HBM from the stem's output through the resnet
    blocks (the blocks' residual stream stays fp32 but is READ as fp16: one matrix-core operand per activation), weights as hi + lo fp16
    parts in registers / LDS (two MFMA products per MAC: the weights keep 22 bits and cost no HBM byte), fp32 accumulation and epilogues.
    Round 4: the TAIL -- the three ConvTranspose2d + BN + ReLU and the head -- keeps fp32 tensors and the 3-term split
    (FFCResNetGenerator.f16_fp32_tail): the oracle with a rounding to fp16 wherever the path has one (O.generator_forward_fp16_storage,
    tools/fp16_by_tensor.py) puts 5-9e-3 max-abs at 1 x 1024^2 on each of the three upsampled tensors alone.  Measured on the GPU at
    4 x 1024^2: 1.68e-2 (round 3) -> 1.19e-2 max-abs, mean-abs 4.0e-4; what is left is the fp16 read of the residual stream by the first
    layer of every block and the bottleneck / front tensors -- removing it means a second activation operand, i.e. the f16x3 path.  Every
    image against the fp32 oracle.  Stated tolerance: **5e-4 mean-abs at every size; max-abs 1e-2 up to 512^2 and 1.5e-2 at 1024^2**
    (round 3: 2e-2), AND no worse than the oracle with the same roundings on the same input (1.5x its max-abs, 1.25x its mean-abs).  The fp32-accurate paths are
    held to 2e-4 above.  Also: the captured graph reproduces the eager result, no range flag."""
    bn, res = shape
    cfg = O.BIG_LAMA
    if 'sd' not in _BIG_SD:
        _BIG_SD['sd'] = O.make_synthetic_state_dict(cfg, seed=0, calib_hw=64)
    gen = make_generator(None, kind='ffc_resnet', **cfg)
    gen.load_state_dict(_BIG_SD['sd'], strict=True)
    gen.cuda().set_precision(L.PREC_F16)
    gen.auto_fallback = False
    x, ref = _oracle_big(bn, res, res, 3000 + res)
    xd = x.cuda()
    y = gen(xd)
    assert gen.precision == L.PREC_F16 and y.dtype == torch.float32
    d = (y.cpu() - ref).abs()
    err = d.amax(dim=(1, 2, 3))
    with torch.no_grad():
        emu = torch.cat([O.generator_forward_fp16_storage(x[i:i + 1], _BIG_SD['sd'], cfg) for i in range(bn)], 0)
    emu_err = (emu - ref).abs().amax(dim=(1, 2, 3))
    print(f'fp16 path {bn} x {res}^2: max-abs {float(err.max()):.2e} mean-abs {float(d.mean()):.2e}; oracle with fp16 storage roundings: {float(emu_err.max()):.2e}')
    assert float(err.max()) < (1e-2 if res <= 512 else 1.5e-2) and float(d.mean()) < 5e-4, (err.tolist(), float(d.mean()))
    # (the maximum is an outlier statistic -- the HIP path's and the oracle's fall on different pixels and images, here 7.4 / 6.4 / 11.9 / 6.6e-3 against
    # 8.4 / 7.6 / 4.5 / 6.0e-3 -- so it is held to 1.5x, the mean to 1.25x)
    assert float(err.max()) < 1.5 * float(emu_err.max()) and float(d.mean()) < 1.25 * float((emu - ref).abs().mean()), (err.tolist(), emu_err.tolist())
    gen.use_graph = True
    yg = gen(xd)
    assert torch.equal(gen(xd), yg) and torch.equal(yg, y)
    if (bn, res) == (4, 1024):      # round 5: the graph of this shape is four parallel parts (or the one-part plan where verify_split found that faster)
        plan = next(iter(gen._plans.values()))
        assert gen._split_parts(xd.shape, xd.device) == plan.get('nsplit', 1) and plan.get('nsplit', 1) in (1, 4)
        gen.split_batch = 4          # ... and forced: the same bits as the eager one-part plan above
        gen._plans.clear()
        assert torch.equal(gen(xd), y)
    gen._plans.clear()


def test_odd_sized_input_generic_fft(big):
    """H, W multiples of 8 only -> bottleneck 21x27 (odd, non power-of-two): generic DFT kernels."""
    cfg, sd, gen, TOL = big
    batch = O.make_synthetic_batch(1, 168, 216, seed=5)
    x = torch.cat([batch['image'] * (1 - batch['mask']), batch['mask']], 1)
    with torch.no_grad():
        ref = O.generator_forward(x, sd, cfg)
    assert float((gen(x.cuda()).cpu() - ref).abs().max()) < TOL


SWEEP_SHAPES = [(1, 16, 16), (1, 16, 24), (1, 32, 64), (2, 72, 104), (3, 136, 136), (1, 200, 328), (2, 312, 120), (1, 400, 408), (1, 512, 520), (1, 520, 512),
                (1, 24, 512), (1, 512, 24), (5, 96, 160), (7, 64, 72), (9, 40, 48), (1, 88, 1048), (2, 1024, 16), (1, 640, 808)]


def test_biglama_shape_sweep(big):
    """Padded shapes real directories produce (any multiple of 8 per side, evaluation/data.py:29-33), chosen to cross the geometry decisions of the
    kernels: bottleneck planes of 2 x 2 ... 80 x 101 (mixed-radix plans incl. primes, planes narrower than a Winograd segment or a 128-pixel tile,
    launches of fewer / more than one round of workgroups, widths that are not multiples of 32 for the stem / head / transposed convs, odd batches).
    Every shape against the fp32 oracle, first call (plan build) and the replay of the captured plan."""
    cfg, sd, gen, TOL = big
    if gen.precision != L.PREC_F16X3:
        pytest.skip('one precision is enough for this sweep')
    for B, H, W in SWEEP_SHAPES:
        batch = O.make_synthetic_batch(B, H, W, seed=H * 7 + W)
        x = torch.cat([batch['image'] * (1 - batch['mask']), batch['mask']], 1)
        with torch.no_grad():
            ref = O.generator_forward(x, sd, cfg)
        xd = x.cuda()
        e1 = float((gen(xd).cpu() - ref).abs().max())
        e2 = float((gen(xd).cpu() - ref).abs().max())
        gen._plans.clear()
        assert e1 < TOL and e2 < TOL, (B, H, W, e1, e2)


def test_photo_sized_input(big):
    """1080 x 1920 (a photo; bottleneck planes 135 x 240: the generic DFT kernels with their one Cooley-Tukey split per length, and 254
    pixel tiles per launch, i.e. conv1 riding in the global-branch epilogues) against the oracle, at 1.5x the 512^2 tolerance like
    the other high-resolution cases."""
    cfg, sd, gen, TOL = big
    if gen.precision != L.PREC_F16X3:
        pytest.skip('one precision is enough for this 2 M-pixel oracle pass')
    x, ref = _oracle_big(1, 1080, 1920, 77)
    y = gen(x.cuda()).cpu()
    gen._plans.clear()
    assert float((y - ref).abs().max()) < 1.5 * TOL, float((y - ref).abs().max())


def test_predict_cli_end_to_end(tmp_path):
    """python -m lama_amd.predict on a checkpoint directory (config.yaml with unresolved interpolations + models/best.ckpt) and a
    folder of PNGs: same on-disk contract as bin/predict.py; results within 1 u8 level of the oracle's batch-1 predict loop."""
    import yaml
    from PIL import Image
    from lama_amd import predict as P
    cfg = O.small_config(ngf=16, n_blocks=2)
    sd = O.make_synthetic_state_dict(cfg, seed=5, calib_hw=32)
    raw = dict(training_model=dict(kind='default', concat_mask=True),
               generator=dict(kind='ffc_resnet', **{k: v for k, v in cfg.items() if not isinstance(v, dict)},
                              init_conv_kwargs=dict(ratio_gin=0, ratio_gout=0, enable_lfu=False),
                              downsample_conv_kwargs=dict(ratio_gin='${generator.init_conv_kwargs.ratio_gout}',
                                                          ratio_gout='${generator.downsample_conv_kwargs.ratio_gin}', enable_lfu=False),
                              resnet_conv_kwargs=dict(ratio_gin=0.75, ratio_gout='${generator.resnet_conv_kwargs.ratio_gin}', enable_lfu=False)))
    mdir = tmp_path / 'model'
    os.makedirs(mdir / 'models')
    with open(mdir / 'config.yaml', 'w') as f:
        yaml.safe_dump(raw, f)
    torch.save({'state_dict': {**{'generator.' + k: v for k, v in sd.items()}, 'discriminator.x': torch.zeros(2)}}, mdir / 'models' / 'best.ckpt')
    rng = np.random.RandomState(3)
    indir = tmp_path / 'in'
    os.makedirs(indir / 'sub')
    shapes = [(100, 136), (128, 128), (100, 136)]
    for i, (h, w) in enumerate(shapes):
        d = indir / ('sub' if i == 2 else '')
        Image.fromarray(rng.randint(0, 256, (h, w, 3)).astype('uint8')).save(d / f'im{i}.png')
        m = np.zeros((h, w), 'uint8')
        m[h // 3: 2 * h // 3, w // 4: w // 2] = 255
        Image.fromarray(m).save(d / f'im{i}_mask001.png')
    out = tmp_path / 'out'
    assert P.main([f'model.path={mdir}', f'indir={indir}', f'outdir={out}', 'batch_size=2']) == 0
    items = P.list_dataset(str(indir) + os.sep, '.png')
    assert len(items) == 3
    sdg = {'generator.' + k: v for k, v in sd.items()}
    for mask_path, img_path in items:
        rel = os.path.splitext(mask_path[len(str(indir)) + 1:])[0] + '.png'
        got = np.array(Image.open(out / rel))
        _, u8 = O.predict_one(O.load_image(img_path, 'RGB'), O.load_image(mask_path, 'L'), sdg, cfg)
        assert got.shape == u8.shape
        assert np.abs(got.astype(int) - u8.astype(int)).max() <= 1, rel
    # dataset.scale_factor (evaluation/data.py:74-77): image (INTER_AREA) and mask (INTER_NEAREST) rescaled on the host, fp32 tensors up, results at
    # the rescaled size (cv2 itself is not in this image: tests/test_scale_factor.py)
    out2 = tmp_path / 'out_scaled'
    assert P.main([f'model.path={mdir}', f'indir={indir}', f'outdir={out2}', 'batch_size=2', 'dataset.scale_factor=0.75']) == 0
    for mask_path, img_path in items:
        rel = os.path.splitext(mask_path[len(str(indir)) + 1:])[0] + '.png'
        got = np.array(Image.open(out2 / rel))
        image = P.scale_image(O.load_image(img_path, 'RGB'), 0.75)
        mask = P.scale_image(O.load_image(mask_path, 'L')[None], 0.75, interpolation='nearest')[0]
        _, u8 = O.predict_one(image, mask, sdg, cfg)
        assert got.shape == u8.shape == (image.shape[1], image.shape[2], 3)
        assert np.abs(got.astype(int) - u8.astype(int)).max() <= 1, rel


def test_long_plane_two_pass_fft(big):
    """1 x 2048 x 256 input -> bottleneck planes 256 x 32: the two-pass LDS FFT (rows, then columns) inside the full generator."""
    cfg, sd, gen, TOL = big
    batch = O.make_synthetic_batch(1, 2048, 256, seed=8)
    x = torch.cat([batch['image'] * (1 - batch['mask']), batch['mask']], 1)
    with torch.no_grad():
        ref = O.generator_forward(x, sd, cfg)
    assert float((gen(x.cuda()).cpu() - ref).abs().max()) < TOL


def test_reference_unit_goldens_on_hardware(prec, golden_dir):
    """VERDICT r4 Next #6 (rows a1-a5 directly evidenced): tests/golden/ffc_units.npz -- FourierUnit at even / odd / prime plane sizes,
    SpectralTransform, FFC_BN_ACT and FFCResnetBlock recorded from the REFERENCE's own classes (make_golden.py) -- replayed through the C ABI
    on the GPU (the CPU twin runs the emulator: test_host_emu.py::test_units_match_golden)."""
    import torch.nn as nn
    from lama_amd import ffc as F
    tol = {L.PREC_F32: 5e-5, L.PREC_F16X3: 5e-5, L.PREC_BF16X3: 1e-3}[prec]
    g = np.load(os.path.join(golden_dir, 'ffc_units.npz'))
    for tag in ('e', 'o', 'p', 'q'):
        sd = {k[len(f'fu_{tag}_sd_'):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(f'fu_{tag}_sd_')}
        c = sd['bn.weight'].numel() // 2
        fu = F.FourierUnit(c, c)
        fu.load_state_dict(sd, strict=True)
        fu.cuda().set_precision(prec)
        y = fu(torch.from_numpy(g[f'fu_{tag}_x']).cuda())
        assert np.abs(y.cpu().numpy() - g[f'fu_{tag}_y']).max() < tol, tag
    blk = F.FFCResnetBlock(16, padding_type='reflect', norm_layer=nn.BatchNorm2d, activation_layer=nn.ReLU,
                           ratio_gin=0.75, ratio_gout=0.75, enable_lfu=False)
    blk.load_state_dict({k[len('blk_sd_'):]: torch.from_numpy(g[k]) for k in g.files if k.startswith('blk_sd_')}, strict=True)
    blk.cuda().set_precision(prec)
    xl, xg = torch.from_numpy(g['blk_xl']).cuda(), torch.from_numpy(g['blk_xg']).cuda()
    yl, yg = blk((xl, xg))
    assert np.abs(yl.cpu().numpy() - g['blk_yl']).max() < tol and np.abs(yg.cpu().numpy() - g['blk_yg']).max() < tol
    l1, g1 = blk.conv1((xl, xg))
    assert np.abs(l1.cpu().numpy() - g['blk_c1_l']).max() < tol and np.abs(g1.cpu().numpy() - g['blk_c1_g']).max() < tol
    st = blk.conv1.ffc.convg2g(xg)
    assert np.abs(st.cpu().numpy() - g['blk_st']).max() < tol
    yl2, yg2 = blk((xl, xg))          # the fused packing is restored after the stand-alone SpectralTransform call
    assert torch.equal(yl2, yl) and torch.equal(yg2, yg)


def test_reference_units_at_biglama_channel_counts_on_hardware(big, golden_dir):
    """... and at the bottleneck shape of BASELINE configs[1]: the reference's FFCResnetBlock(512) = (128 | 384) channels on [2, 512, 64, 64]
    (tests/golden/make_golden_units512.py) -- block, FFC_BN_ACT, SpectralTransform, FourierUnit as stand-alone calls of generator.model[5]:
    the Winograd local conv, the 12 x 1 global launch with the fused conv1, gemm1x1_wk and the one-buffer 64 x 64 FFTs against the reference."""
    import sys
    sys.path.insert(0, golden_dir)
    from make_golden_units512 import BLOCK, block_inputs, sample
    cfg, sd, gen, TOL = big
    g = np.load(os.path.join(golden_dir, 'ffc_block512.npz'))
    blk = gen.model[int(BLOCK.split('.')[1])]
    xl, xg = (t.cuda() for t in block_inputs())
    rel = {L.PREC_F32: 2e-5, L.PREC_F16X3: 2e-5, L.PREC_BF16X3: 4e-4}[gen.precision]
    yl, yg = blk((xl, xg))
    l1, g1 = blk.conv1((xl, xg))
    st = blk.conv1.ffc.convg2g(xg)
    fu = blk.conv1.ffc.convg2g.fu(xg[:, :192].contiguous())
    for a, key in ((yl, 'yl'), (yg, 'yg'), (l1, 'c1_l'), (g1, 'c1_g'), (st, 'st'), (fu, 'fu')):
        err = np.abs(sample(a.float().cpu()) - g[key + '_sample']).max()
        assert err < rel * max(1.0, float(g[key + '_stat'][2])), (key, err)
    y2 = gen.model[5:7]((xl, xg))      # two blocks as a slice: the stand-alone calls above left the fused packing intact
    assert torch.isfinite(y2[0]).all() and torch.isfinite(y2[1]).all()


def test_host_fed_step_graph_matches_plain_forward():
    """lama_amd.predict.HostFedStep on hardware: upload of batch k + 1, compute of batch k and download of batch k - 1 per launch -- on copy streams
    beside plain launches (the default) and as parallel branches of ONE captured hipGraph per step (VERDICT r4 Next #4; measured slower) -- six
    steps with different pinned-host inputs per step deliver, in order and bit for bit, the u8 images of the plain (generator-graph) forward on
    each batch; the multi-rank form (no download branch) leaves them in u8[p]."""
    from lama_amd.predict import HostFedStep
    cfg = O.small_config(ngf=16, n_blocks=2)
    sd = {'generator.' + k: v for k, v in O.make_synthetic_state_dict(cfg, seed=3, calib_hw=32).items()}
    model = trainers.DefaultInpaintingTrainingModule(dict(generator=dict(kind='ffc_resnet', **cfg)))
    model.load_state_dict(sd, strict=True)
    model.freeze().cuda()
    model.generator.use_graph = True
    lib = model.generator._exec.lib
    n, H, W, steps = 2, 128, 160, 6
    batches = [O.make_synthetic_batch(n, H, W, seed=40 + k) for k in range(steps)]
    want = []
    for b in batches:
        out = model(dict(image=b['image'].cuda(), mask=(b['mask'].cuda() > 0) * 1))['inpainted']
        u8 = torch.empty(n, H, W, 3, dtype=torch.uint8, device='cuda')
        lib.quantize_u8_hwc(L.view(out), u8, n, H, W, torch.cuda.current_stream().cuda_stream)
        want.append(u8.cpu())
    # round 6: the step is fed with u8 HWC images as they are on disk (u8_input, the default) -- here with an UNPADDED 125 x 157 image in each
    # 128 x 160 slot: / 255, the symmetric padding and mask > 0 run on the device and must equal the host's (O.load_image / pad_img_to_modulo)
    hv, wv = H - 3, W - 3
    raw = [dict(image=(b['image'][:, :, :hv, :wv] * 255).round().to(torch.uint8).permute(0, 2, 3, 1).contiguous().numpy(),
                mask=(b['mask'][:, 0, :hv, :wv] * 255).round().to(torch.uint8).numpy()) for b in batches]
    want_u8 = []
    for r in raw:
        im = np.stack([O.pad_img_to_modulo(np.transpose(x, (2, 0, 1)).astype('float32') / 255, 8) for x in r['image']])
        mk = np.stack([O.pad_img_to_modulo(x[None].astype('float32') / 255, 8) for x in r['mask']])
        out = model(dict(image=torch.from_numpy(im).cuda(), mask=(torch.from_numpy(mk).cuda() > 0) * 1))['inpainted']
        u8 = torch.empty(n, H, W, 3, dtype=torch.uint8, device='cuda')
        lib.quantize_u8_hwc(L.view(out), u8, n, H, W, torch.cuda.current_stream().cuda_stream)
        want_u8.append(u8.cpu())
    for drain, mode, u8_in in ((True, 'replay', False), (False, 'replay', False), (True, 'streams', False), (True, 'graph', False), (False, 'graph', False),
                               (True, 'replay', True), (True, 'graph', True), (False, 'graph', True), (True, 'host', True), (False, 'host', True), (True, 'host', False)):
        hs = HostFedStep(model, n, H, W, 'cuda', drain=drain, mode=mode, u8_input=u8_in)
        assert hs.u8_input == u8_in

        def fill(p, k):
            if u8_in:
                for j in range(n):
                    hs.put(p, j, raw[k]['image'][j], raw[k]['mask'][j])
                return
            im, mk = hs.host(p)
            im[:] = batches[k]['image'].numpy()
            mk[:] = batches[k]['mask'].numpy()

        got = {}
        fill(0, 0)
        hs.prime(0)
        for k in range(steps):
            p = k & 1
            if k + 1 < steps:
                fill(1 - p, k + 1)
            hs.launch(p)
            hs.wait(p)
            if not drain:
                got[k] = hs.u8[p].cpu()
            elif k >= 1:
                got[k - 1] = torch.from_numpy(hs.result(1 - p).copy())
        hs.flush((steps - 1) & 1)
        if drain:
            got[steps - 1] = torch.from_numpy(hs.result((steps - 1) & 1).copy())
        assert sorted(got) == list(range(steps))
        for k in range(steps):
            assert torch.equal(got[k], (want_u8 if u8_in else want)[k]), (drain, mode, u8_in, k)
        assert (hs.graphs[0] is not None and hs.graphs[1] is not None) == (mode == 'graph')
    assert model.generator.check_range('cuda') is True and model.generator.use_graph is True


def test_input_buffer_and_unclone_are_the_same_forward(big):
    """VERDICT r4 Next #7: mask_compose writes straight into the plan's static input (generator.input_buffer) and blend reads the plan's output
    (clone_output False via keep_predicted_image False): same bits as the copying path, and the default still returns tensors of its own."""
    cfg, sd, gen, TOL = big
    model = trainers.DefaultInpaintingTrainingModule(dict(generator=dict(kind='ffc_resnet', **cfg)))
    model.generator = gen
    gen.use_graph = True
    try:
        b = O.make_synthetic_batch(2, 256, 256, seed=5)
        img, mask = b['image'].cuda(), b['mask'].cuda()
        o1 = model(dict(image=img, mask=mask))
        p1, i1 = o1['predicted_image'], o1['inpainted'].clone()
        masked = torch.cat([img * (1 - mask), mask], 1)
        assert torch.equal(gen(masked), p1)                                   # a caller's own tensor: staged, same result
        model.keep_predicted_image = False
        o2 = model(dict(image=img, mask=mask))
        assert torch.equal(o2['inpainted'], i1) and torch.equal(o2['predicted_image'], p1)
        plan = gen._plans[((2, 4, 256, 256), str(img.device))]
        assert o2['predicted_image'].data_ptr() == plan['static_out'].data_ptr() and p1.data_ptr() != plan['static_out'].data_ptr()
        assert gen.input_buffer((2, 4, 256, 256), img.device).data_ptr() == plan['static_in'].data_ptr()
        assert gen.clone_output is True
    finally:
        gen.use_graph = False
        gen._plans.clear()


def test_split_batch_graph_is_bit_identical(big):
    """Round 5: 8 x 512^2 as two (auto) / four parallel branches of one captured hipGraph -- the parts' launches carry LAMA_CONV_SIBLINGS_* and take
    the kernel geometry of the whole batch, so the output equals the one-part plan's BIT FOR BIT, eager and replayed, twice (dirty buffers)."""
    cfg, sd, gen, TOL = big
    batch = O.make_synthetic_batch(8, 512, 512, seed=43)
    xd = torch.cat([batch['image'] * (1 - batch['mask']), batch['mask']], 1).cuda()
    assert gen._split_parts(xd.shape, xd.device) == 1                       # plain launches: never split
    gen.use_graph = True
    gen._split_ok.clear()                                                    # (an earlier test's verify_split verdict on this box: the RULE is asserted here)
    auto = gen._split_parts(xd.shape, xd.device)
    assert auto == (4 if gen.precision in (L.PREC_F16X3, L.PREC_BF16X3) else 1)
    assert gen._split_parts((4, 4, 1024, 1024), xd.device) == auto and gen._split_parts((4, 4, 512, 512), xd.device) == 1
    assert gen._split_parts((6, 4, 512, 512), xd.device) == 1 and gen._split_parts((2, 4, 1024, 1024), xd.device) == (2 if auto == 4 else 1)
    try:
        gen.split_batch = 1
        ref = gen(xd).clone()
        for n in (2, 4):
            gen.split_batch = n
            for graph in (False, True):
                gen.use_graph = graph
                gen._plans.clear()
                if auto == 4:
                    assert torch.equal(gen(xd), ref) and torch.equal(gen(xd), ref), (n, graph)
                else:   # exact fp32: the one-part plan runs the local conv beside the spectral branch in its cooperative geometry, the parts are
                    #     one-stream plans (another fp32 summation order)
                    assert float((gen(xd) - ref).abs().max()) < 1e-5 and float((gen(xd) - ref).abs().max()) < 1e-5, (n, graph)
                assert next(iter(gen._plans.values()))['nsplit'] == n
    finally:
        gen.split_batch = None
        gen.use_graph = False
        gen._plans.clear()
    assert gen.check_range(xd.device) is True


def test_verify_split_keeps_the_faster_plan(big, monkeypatch):
    """generator.verify_split: the split plan the rule proposes is timed ONCE per shape against the one-part plan and kept only if it is not slower.
    On a runtime that runs kernel branches side by side the two are within a few per cent and the faster one stays (the split plan on most
    boxes); where the branches are serialised (simulated here by a 5 ms stall at the head of the branches -- rocprofv3's kernel trace does it
    for real) the generator falls back to the one-part plan.  Same bits either way."""
    cfg, sd, gen, TOL = big
    if gen.precision not in (L.PREC_F16X3, L.PREC_BF16X3):
        pytest.skip('the rule splits the split precisions only')
    batch = O.make_synthetic_batch(8, 512, 512, seed=47)
    xd = torch.cat([batch['image'] * (1 - batch['mask']), batch['mask']], 1).cuda()
    key = gen._plan_key(xd.shape, xd.device)
    assert key == gen._plan_key(xd.shape, 'cuda')                           # 'cuda' and 'cuda:0' are one plan-cache key
    try:
        gen.use_graph = True
        gen._plans.clear(); gen._split_ok.clear()
        y = gen(xd).clone()
        t = gen.split_decisions[key]
        print('verify_split:', t, flush=True)
        # the split plan stays only when its median is at least split_margin (2 %) below the one-part plan's: a tie keeps the one-part plan
        assert t['parts'] == 4 and t['kept'] == (4 if t['ms_split'] <= (1 - gen.split_margin) * t['ms_one_part'] else 1)
        assert len(gen._plans) == 1                                        # the loser was released
        assert gen._split_parts(xd.shape, xd.device) == t['kept'] and gen._plans[key].get('nsplit', 1) == t['kept']
        assert t['ms_split'] < 1.15 * t['ms_one_part'], t          # ... and on a healthy runtime the two are within a few per cent
        assert torch.equal(gen(xd), y)
        # a runtime that stalls every branch: the check must reject the split plan
        real = type(gen)._run_split

        def stalled(self, plan, x):
            torch.cuda._sleep(12_000_000)
            return real(self, plan, x)

        monkeypatch.setattr(type(gen), '_run_split', stalled)
        gen._plans.clear(); gen._split_ok.clear()
        y1 = gen(xd).clone()
        t = gen.split_decisions[key]
        assert t['kept'] == 1 and gen._split_parts(xd.shape, xd.device) == 1 and 'parts' not in gen._plans[key]
        assert torch.equal(y1, y)
    finally:
        gen.use_graph = False
        gen._plans.clear(); gen._split_ok.clear()
