"""GPU parity of the refinement path (BASELINE configs[4], saicinpainting/evaluation/refinement.py) against the CPU oracle:
gradients of the explicit reverse pass at big-lama's channel counts against torch autograd, and refine_predict end to end."""
import os

import numpy as np
import pytest
import torch

from lama_amd import _lib as L
from lama_amd import ffc as F
from lama_amd import refinement as RF
from lama_amd import trainers
from lama_amd.backward import RearPass
from lama_amd.modules import make_generator
from oracle import lama_oracle as O
from oracle import refine_oracle as R

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.fixture(scope='module')
def big2():
    """big-lama channel counts (512-channel bottleneck, 128 | 384 split, 192-channel spectral branch), two resnet blocks."""
    cfg = dict(O.BIG_LAMA)
    cfg['n_blocks'] = 2
    sd = O.make_synthetic_state_dict(cfg, seed=2, calib_hw=64)
    return cfg, sd


@pytest.mark.parametrize('res', [64, 256], ids=['64px_strict', '256px_planes32'])
@pytest.mark.parametrize('fwd_prec,bwd_prec,tol', [(L.PREC_F32, L.PREC_F32, 5e-5), (L.PREC_F16X3, L.PREC_BF16X3, 3e-3)],
                         ids=['f32', 'f16x3_fwd_bf16x3_bwd'])
def test_rear_gradients_full_channel_count(big2, fwd_prec, bwd_prec, tol, res):
    """d loss / d (z1, z2) through two FFCResnetBlocks (incl. the FourierUnit adjoints), the three fused ConvTranspose2d + BN + ReLU
    and the 7x7 head + sigmoid at big-lama's channel counts, against torch autograd through the oracle.

    ReLU' is discontinuous: a pre-activation within rounding distance of 0 gets mask 1 on one side and 0 on the other (two valid
    fp32 evaluations), and each such flip changes the gradient by 100 % of one element, which the dgrad convs spread over a
    neighbourhood and the FourierUnit adjoint over a whole plane.  At 64 x 64 (0.3 M ReLU outputs) no flip occurs and the
    comparison is strict; at 256 x 256 (7 M ReLU outputs, a handful of flips: tools/dbg_bwd.py) the criterion is the relative L2
    error plus the fraction of elements that are off by more than 5 % of the largest gradient."""
    cfg, sd = big2
    gen = make_generator(None, kind='ffc_resnet', **cfg)
    gen.load_state_dict(sd, strict=True)
    gen.cuda().set_precision(fwd_prec)
    fri = R.first_resblock_index(cfg)
    batch = O.make_synthetic_batch(1, res, res, seed=6)
    x = torch.cat([batch['image'] * (1 - batch['mask']), batch['mask']], 1)
    with torch.no_grad():
        z1, z2 = O.run_layers(x, sd, cfg, 0, fri)
    z1r, z2r = z1.clone().requires_grad_(True), z2.clone().requires_grad_(True)
    pred_ref = O.run_layers((z1r, z2r), sd, cfg, fri, None)
    gw = torch.randn(pred_ref.shape, generator=torch.Generator().manual_seed(7)) / pred_ref.numel()
    (pred_ref * gw).sum().backward()
    gref = torch.cat([z1r.grad, z2r.grad], 1)
    rear = RearPass(gen, fri, bwd_precision=bwd_prec)
    pred = rear.forward(torch.cat([z1, z2], 1).contiguous().to(DEV))
    assert float((pred.cpu() - pred_ref.detach()).abs().max()) < 2e-4
    g = rear.backward(gw.contiguous().to(DEV)).cpu()
    d = (g - gref).abs()
    gmax = float(gref.abs().max())
    if res == 64:
        assert float(d.max()) / gmax < tol, float(d.max()) / gmax
    else:
        l2 = float(d.norm() / gref.norm())
        frac = float((d > 0.05 * gmax).float().mean())
        assert l2 < 2e-2 and frac < 1e-3, (l2, frac, float(d.max()) / gmax)


def test_refine_predict_end_to_end():
    """refine_predict on a 2-scale pyramid (small generator, fp32 forward, bf16x3 reverse pass) against the oracle's
    torch-autograd + Adam loop: same losses per iteration, same inpainting."""
    cfg = O.small_config(ngf=16, n_blocks=2)
    sd = O.make_synthetic_state_dict(cfg, seed=21, calib_hw=32)
    model = trainers.DefaultInpaintingTrainingModule(dict(generator=dict(kind='ffc_resnet', **cfg)))
    model.load_state_dict({'generator.' + k: v for k, v in sd.items()}, strict=True)
    model.freeze().cuda()
    g = torch.Generator().manual_seed(5)
    Hh, Ww = 250, 300
    image = torch.rand(1, 3, 256, 304, generator=g)
    mask = torch.zeros(1, 1, 256, 304)
    mask[:, :, 40:200, 50:250] = 1.0
    batch = dict(image=image.cuda(), mask=mask.cuda(), unpad_to_size=[torch.tensor([Hh]), torch.tensor([Ww])])
    trace = []
    out = RF.refine_predict(batch, model, gpu_ids='0,', modulo=8, n_iters=5, lr=0.002, min_side=125, max_scales=2, px_budget=10 ** 7,
                            trace=trace)
    ref = R.refine_predict(image, mask, (Hh, Ww), sd, cfg, modulo=8, n_iters=5, lr=0.002, min_side=125, max_scales=2, px_budget=10 ** 7)
    assert out.shape == ref.shape == (1, 3, Hh, Ww) and len(trace) == 2 and len(trace[1]['loss']) == 5
    assert float((out - ref).abs().max()) < 1e-2 and float((out - ref).abs().mean()) < 3e-4
    assert trace[1]['loss'][-1] < trace[1]['loss'][0]           # the refinement does reduce its loss


@pytest.fixture(scope='module')
def biglama_module():
    """The FULL big-lama generator (18 FFCResnetBlocks) with the seeded synthetic weights of the goldens (seed 0, calib 64)."""
    cfg = dict(O.BIG_LAMA)
    sd = O.make_synthetic_state_dict(cfg, seed=0, calib_hw=64)
    model = trainers.DefaultInpaintingTrainingModule(dict(generator=dict(kind='ffc_resnet', **cfg)))
    model.load_state_dict({'generator.' + k: v for k, v in sd.items()}, strict=True)
    model.freeze().cuda()
    checksum = sum(float(v.double().sum()) for v in sd.values() if v.is_floating_point())
    return model, checksum


@pytest.mark.parametrize('precs', [(L.PREC_F16X3, None), (L.PREC_F32, L.PREC_F32)], ids=['default_f16x3_fwd_bf16x3_bwd', 'exact_f32'])
@pytest.mark.parametrize('res', [1024, 2048], ids=['2scales_512_1024', 'configs4_3scales_2048'])
def test_refine_predict_biglama_golden(biglama_module, golden_dir, res, precs):
    """BASELINE configs[4] at its own configuration: ``refine_predict`` on the full big-lama generator, 15 iterations per scale
    (res = 2048: exactly what bench.py's ``configs4_refine_leg`` times -- 3 scales 512 / 1024 / 2048, px_budget 4194304), against
    the loss curves and output samples that ``tests/golden/make_golden_refine.py`` recorded from the CPU oracle (torch autograd +
    torch.optim.Adam through all 18 blocks), in the default precision (f16x3 forward, bf16x3 reverse pass) and in exact fp32.

    What can be asked of the OUTPUT.  The refinement loss is an L1: its gradient is sign(pred - image), so a 1e-5 difference in the
    prediction flips the sign for ~0.1 % of the pixels (measured on the GPU box, tools/refine_diag.py: first-iteration gradient of the
    EXACT-fp32 reverse pass 1.1e-2 relative L2 from autograd's with pred0 4e-5 apart), and Adam turns a sign change of a gradient element
    into a full learning-rate step the other way: the 15-iteration trajectory is chaotic in the rounding of the forward pass.  The
    golden file therefore carries the algorithm's OWN sensitivity: the oracle re-run with the initial features perturbed by a relative
    2e-6 (two valid fp32 evaluations of the front layers differ by that much) lands 1.14e-3 mean-abs / 3.3e-2 max-abs from the golden
    output at the 1024 scale while its loss curve stays within 9e-6 -- and the HIP path lands 1.15e-3 / 3.3e-2 / 1.1e-5 from it, in
    exact fp32 and in the default precision alike (profiles/r03_refine_parity.txt).  Bars: every loss within max(1e-4, 4x the self-sensitivity) of the oracle's and monotonically falling; the refined image within 2x the oracle's self-distance (mean and max; floors
    3e-4 / 5e-3, which the first scale -- a plain forward -- meets by three orders of magnitude); and clearly different from the
    un-refined forward (so "did nothing" cannot pass)."""
    path = os.path.join(golden_dir, f'refine_biglama_{res}.npz')
    if not os.path.exists(path):
        pytest.skip(f'{path} not generated')
    g = np.load(path)
    model, checksum = biglama_module
    assert abs(checksum - float(g['sd_checksum'][0])) < 1e-3 * abs(checksum), 'seeded weights drifted from the golden run'
    import importlib.util
    spec = importlib.util.spec_from_file_location('make_golden_refine', os.path.join(golden_dir, 'make_golden_refine.py'))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    image, mask = mk.make_case(res=res, seed=int(g['seed_img'][0]))
    n_scales, px_budget = (2, 1800000) if res <= 1024 else (3, 4194304)
    batch = dict(image=image.cuda(), mask=mask.cuda(), unpad_to_size=[torch.tensor([res]), torch.tensor([res])])
    trace = []
    model.generator.set_precision(precs[0])
    try:
        out = RF.refine_predict(batch, model, gpu_ids='0,', modulo=8, n_iters=int(g['n_iters'][0]), lr=0.002, min_side=512,
                                max_scales=n_scales, px_budget=px_budget, trace=trace, bwd_precision=precs[1])
    finally:
        model.generator.set_precision(L.PREC_F16X3)
    assert out.shape == (1, 3, res, res) and len(trace) == n_scales
    report = []
    for s in range(n_scales):
        ref_loss = g[f'loss{s}']
        got = np.asarray(trace[s].get('loss', []), dtype=np.float64)
        assert got.shape == ref_loss.shape
        if len(ref_loss):
            rel = np.abs(got - ref_loss) / ref_loss
            report.append(f'scale {s}: max loss rel err {rel.max():.2e}')
            self_rel = float(g[f'self{s}'][2]) if f'self{s}' in g.files else 0.0
            assert rel.max() < max(1e-4, 4 * self_rel), (s, got, ref_loss)
            assert np.all(np.diff(got) < 0)
        o = trace[s]['out']
        st = o.shape[-1] // g[f'out{s}_sample'].shape[-1]
        d = (o[:, :, ::st, ::st].numpy() - g[f'out{s}_sample'])
        report.append(f'scale {s}: out mean-abs {np.abs(d).mean():.2e} max-abs {np.abs(d).max():.2e}')
        self_mean, self_max = (float(g[f'self{s}'][0]), float(g[f'self{s}'][1])) if f'self{s}' in g.files else (0.0, 0.0)
        report.append(f'         (oracle vs perturbed oracle: mean-abs {self_mean:.2e} max-abs {self_max:.2e})')
        print('\n'.join(report[-2:]), flush=True)
        assert np.abs(d).mean() < max(3e-4, 2 * self_mean) and np.abs(d).max() < max(5e-3, 2 * self_max), report
    st = int(g['sample_stride'][0])
    moved = float(np.abs(out[:, :, ::st, ::st].numpy() - g['plain_sample']).mean())
    assert moved > 0.3 * float(g['refine_minus_plain_meanabs'][0]), (moved, g['refine_minus_plain_meanabs'])
    print('\n'.join(report))


@pytest.mark.parametrize('precs', [(L.PREC_F32, L.PREC_F32), (L.PREC_F16X3, None)], ids=['exact_f32', 'default_f16x3_fwd_bf16x3_bwd'])
def test_first_iteration_gradient_18_blocks(biglama_module, golden_dir, precs):
    """The FIRST refinement iteration through ALL 18 FFCResnetBlocks at 512 x 512 (refinement.py:131-165): the prediction ``pred0`` and the
    gradient that ``loss.backward()`` leaves on (z1, z2), against the committed sample of the CPU oracle (torch autograd;
    tests/golden/make_golden_refine_grad.py).  Holds the explicit reverse pass (lama_amd/backward.py) at full depth -- 36 FourierUnit adjoints,
    36 x 3 dgrad 3x3 convs, three ConvTranspose2d adjoints and the head -- where ``test_rear_gradients_full_channel_count`` has two blocks.

    Bars.  pred0: relative L2 <= 1e-4 (exact fp32; measured ~2e-6) / 3e-4 (f16x3 forward).  Gradient: the loss is an L1, so its derivative with
    respect to the prediction is sign(pred - image) / N and every pixel whose prediction lies within rounding distance of its target flips; the
    golden file carries the oracle's OWN distance to itself under a relative 2e-6 perturbation of z (1.16e-2 relative L2) and the HIP gradient
    must stay within 2x that of the golden -- and correlate with it to better than 0.999 (a wrong adjoint anywhere in the 18 blocks does not)."""
    path = os.path.join(golden_dir, 'refine_grad_biglama_512.npz')
    g = np.load(path)
    model, checksum = biglama_module
    assert abs(checksum - float(g['sd_checksum'][0])) < 1e-3 * abs(checksum), 'seeded weights drifted from the golden run'
    import importlib.util
    spec = importlib.util.spec_from_file_location('make_golden_refine', os.path.join(golden_dir, 'make_golden_refine.py'))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    image, mask = mk.make_case(res=512, seed=int(g['seed_img'][0]))
    batch = dict(image=image.cuda(), mask=mask.cuda(), unpad_to_size=[torch.tensor([512]), torch.tensor([512])])
    trace = []
    model.generator.set_precision(precs[0])
    try:
        RF.refine_predict(batch, model, gpu_ids='0,', modulo=8, n_iters=2, lr=0.002, min_side=256, max_scales=2, px_budget=10 ** 8,
                          trace=trace, bwd_precision=precs[1])
    finally:
        model.generator.set_precision(L.PREC_F16X3)
    assert len(trace) == 2 and len(trace[1]['loss']) == 2
    gs, ps = int(g['strides'][0]), int(g['strides'][1])
    pred0 = trace[1]['pred0'].cpu()[:, :, ::ps, ::ps].numpy()
    gz = trace[1]['g_z'].cpu()
    assert gz.shape == (1, 512, 64, 64)
    rel_p = float(np.linalg.norm(pred0 - g['pred0_sample']) / np.linalg.norm(g['pred0_sample']))
    got, ref = gz[:, :, ::gs, ::gs].numpy().astype(np.float64), g['g_sample'].astype(np.float64)
    rel_g = float(np.linalg.norm(got - ref) / np.linalg.norm(ref))
    corr = float((got * ref).sum() / (np.linalg.norm(got) * np.linalg.norm(ref)))
    norm_ratio = float(gz.double().norm()) / float(g['g_norm'][0])
    loss_rel = np.abs(np.asarray(trace[1]['loss']) - g['loss']) / g['loss']
    print(f'pred0 rel L2 {rel_p:.2e}; gradient rel L2 {rel_g:.2e} (oracle vs itself {float(g["self_rel"][1]):.2e}), corr {corr:.5f}, '
          f'|g| ratio {norm_ratio:.4f}; loss rel {loss_rel.max():.1e}', flush=True)
    assert rel_p < (1e-4 if precs[0] == L.PREC_F32 else 3e-4), rel_p
    assert loss_rel.max() < 1e-4, loss_rel
    assert rel_g < 2.0 * float(g['self_rel'][1]), (rel_g, g['self_rel'])
    assert corr > 0.999 and abs(norm_ratio - 1.0) < 5e-3, (corr, norm_ratio)


@pytest.mark.parametrize('fwd_prec,bwd_prec,bar', [(L.PREC_F32, L.PREC_F32, 5e-5), (L.PREC_F16X3, L.PREC_BF16X3, 3e-4)],
                         ids=['exact_f32', 'default_f16x3_fwd_bf16x3_bwd'])
def test_rear_gradients_18_blocks_strict(biglama_module, fwd_prec, bwd_prec, bar):
    """VERDICT r4 Next #5 -- a check of the reverse pass at FULL depth that depends neither on the optimisation trajectory nor on ReLU-mask
    flips: d <gw, pred> / d (z1, z2) through all 18 FFCResnetBlocks (36 FourierUnit adjoints, 108 dgrad 3x3 convs), the three ConvTranspose2d
    adjoints and the head at 256 x 256 (32 x 32 planes: the Winograd local conv in the tape's forward and the Winograd-interior + frame dgrad in
    the reverse pass), for a fixed random functional gw instead of the L1 loss, against torch autograd through the oracle WITH THE RELU MASKS OF
    THE HIP TAPE (tests/masked_oracle.py).  Without fixed masks the oracle is 5.3e-3 relative L2 from itself under a 1e-7 perturbation of z
    (measured, 128^2 and 256^2: two valid fp32 evaluations flip a handful of the 2e7 ReLUs) -- a bar a 0.5 % systematic error would pass; with
    them only rounding separates the two gradients.  Measured on the MI355X (gpurun_out/r05b): 2.6e-6 relative L2 (max / gmax 3.4e-6) in exact
    fp32, 2.4e-5 with the default precisions (f16x3 forward, bf16x3 reverse) -- and 8.7e-3 / 1.2e-2 against autograd with its OWN masks, which
    is the flip noise the other gradient tests have to allow.  Bars: 5e-5 / 3e-4 relative L2 and 5x that for the largest element."""
    from tests import masked_oracle as MO
    model, _ = biglama_module
    gen = model.generator
    cfg = dict(O.BIG_LAMA)
    sd = {k[len('generator.'):]: v.detach().cpu() for k, v in model.state_dict().items()}
    fri = R.first_resblock_index(cfg)
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
    batch = O.make_synthetic_batch(1, 256, 256, seed=6)
    x = torch.cat([batch['image'] * (1 - batch['mask']), batch['mask']], 1)
    with torch.no_grad():
        z1, z2 = O.run_layers(x, sd, cfg, 0, fri)
    gen.set_precision(fwd_prec)
    try:
        rear = RearPass(gen, fri, bwd_precision=bwd_prec)
        pred = rear.forward(torch.cat([z1, z2], 1).contiguous().to(DEV))
        gw = torch.randn(pred.shape, generator=torch.Generator().manual_seed(7)) / pred.numel()
        g = rear.backward(gw.contiguous().to(DEV)).cpu()
        masks = MO.tape_masks(rear)
    finally:
        gen.set_precision(L.PREC_F16X3)
    assert len(masks) == 18 * 2 * 4 + 3
    pred_ref, gref = MO.rear_gradient(z1, z2, sd, cfg, fri, gw, masks)
    # the same autograd pass with the oracle's OWN masks: how far mask flips alone move the gradient
    z1r, z2r = z1.clone().requires_grad_(True), z2.clone().requires_grad_(True)
    (O.run_layers((z1r, z2r), sd, cfg, fri, None) * gw).sum().backward()
    gown = torch.cat([z1r.grad, z2r.grad], 1)
    rel = float((g - gref).norm() / gref.norm())
    mx = float((g - gref).abs().max() / gref.abs().max())
    rel_own = float((g - gown).norm() / gown.norm())
    perr = float((pred.cpu() - pred_ref).abs().max())
    print(f'18 blocks, 256^2: gradient rel L2 {rel:.2e} (max / gmax {mx:.2e}) against autograd with the tape\'s masks; {rel_own:.2e} against autograd '
          f'with its own masks; pred max-abs {perr:.2e}', flush=True)
    assert perr < (1e-4 if fwd_prec == L.PREC_F32 else 5e-4), perr
    assert rel < bar and mx < 5 * bar, (rel, mx)


def test_rear_gradients_strict_on_256x256_planes(big2):
    """the strict fixed-mask gradient check ran at 256^2 only, so the masked transforms of the reverse pass
    (lama_rfft2_masked_fwd / lama_irfft2_masked_fwd: the two ReLU derivatives of the spectral branch inside the 256 x 256-plane two-pass FFT
    kernels, v108) were pinned by the trajectory goldens and a kernel-level test only.  Here: d <gw, pred> / d (z1, z2) through two
    FFCResnetBlocks at big-lama's channel counts, the three ConvTranspose2d adjoints and the head at 2048 x 2048 -- bottleneck planes of
    256 x 256, BASELINE configs[4]'s top scale -- against torch autograd through the oracle with the ReLU masks of the HIP tape
    (tests/masked_oracle.py): only rounding separates the two.  Exact fp32 both ways; bars as test_rear_gradients_18_blocks_strict."""
    from tests import masked_oracle as MO
    cfg, sd = big2
    gen = make_generator(None, kind='ffc_resnet', **cfg)
    gen.load_state_dict(sd, strict=True)
    gen.cuda().set_precision(L.PREC_F32)
    fri = R.first_resblock_index(cfg)
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
    batch = O.make_synthetic_batch(1, 2048, 2048, seed=9)
    x = torch.cat([batch['image'] * (1 - batch['mask']), batch['mask']], 1)
    with torch.no_grad():
        z1, z2 = O.run_layers(x, sd, cfg, 0, fri)
    assert tuple(z1.shape[-2:]) == (256, 256)
    rear = RearPass(gen, fri, bwd_precision=L.PREC_F32)
    pred = rear.forward(torch.cat([z1, z2], 1).contiguous().to(DEV))
    gw = torch.randn(pred.shape, generator=torch.Generator().manual_seed(7)) / pred.numel()
    g = rear.backward(gw.contiguous().to(DEV)).cpu()
    assert rear._plan['sh'].get('fft_mask', True) is True              # the masked-transform entries took the launch (not the act_bwd fallback)
    masks = MO.tape_masks(rear)
    assert len(masks) == 2 * 2 * 4 + 3
    pred_ref, gref = MO.rear_gradient(z1, z2, sd, cfg, fri, gw, masks)
    rel = float((g - gref).norm() / gref.norm())
    mx = float((g - gref).abs().max() / gref.abs().max())
    perr = float((pred.cpu() - pred_ref).abs().max())
    print(f'2 blocks, 2048^2 (256 x 256 planes, masked FFT entries): gradient rel L2 {rel:.2e} (max / gmax {mx:.2e}), pred max-abs {perr:.2e}', flush=True)
    assert perr < 1e-4, perr
    assert rel < 5e-5 and mx < 2.5e-4, (rel, mx)


@pytest.mark.parametrize('hw', [(1344, 1344), (1000, 1504)], ids=['planes168x168_default_budget', 'planes125x188_prime47'])
def test_rear_gradients_strict_on_non_power_of_two_planes(big2, hw):
    """The reverse pass where the reference's DEFAULT refinement runs it (configs/prediction/default.yaml:24: px_budget 1.8 M rescales a large image to
    ~1341^2 -> padded 1344^2, bottleneck planes 168 x 168 = 2^3 3 7; evaluation/refinement.py:203-211), and at a photo plane with a large prime
    (1000 x 1504 -> 125 x 188 = 5^3 x 4 47): the adjoints of the mixed-radix transforms (fft_mr_dev.inc), the dgrad convs at planes that are no
    multiple of a tile, the transposed-conv adjoints at ragged widths.  Same construction as test_rear_gradients_strict_on_256x256_planes: two
    FFCResnetBlocks at big-lama's channel counts + the three ConvTranspose2d adjoints + the head, exact fp32 both ways, torch autograd through the
    oracle with the ReLU masks of the HIP tape -- only rounding separates the two."""
    from tests import masked_oracle as MO
    cfg, sd = big2
    gen = make_generator(None, kind='ffc_resnet', **cfg)
    gen.load_state_dict(sd, strict=True)
    gen.cuda().set_precision(L.PREC_F32)
    fri = R.first_resblock_index(cfg)
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
    H, W = hw
    batch = O.make_synthetic_batch(1, H, W, seed=10)
    x = torch.cat([batch['image'] * (1 - batch['mask']), batch['mask']], 1)
    with torch.no_grad():
        z1, z2 = O.run_layers(x, sd, cfg, 0, fri)
    assert tuple(z1.shape[-2:]) == (H // 8, W // 8)
    rear = RearPass(gen, fri, bwd_precision=L.PREC_F32)
    pred = rear.forward(torch.cat([z1, z2], 1).contiguous().to(DEV))
    gw = torch.randn(pred.shape, generator=torch.Generator().manual_seed(7)) / pred.numel()
    g = rear.backward(gw.contiguous().to(DEV)).cpu()
    masks = MO.tape_masks(rear)
    assert len(masks) == 2 * 2 * 4 + 3
    pred_ref, gref = MO.rear_gradient(z1, z2, sd, cfg, fri, gw, masks)
    rel = float((g - gref).norm() / gref.norm())
    mx = float((g - gref).abs().max() / gref.abs().max())
    perr = float((pred.cpu() - pred_ref).abs().max())
    print(f'2 blocks, {H} x {W} (planes {H // 8} x {W // 8}; masked FFT entries: {rear._plan["sh"].get("fft_mask", True)}): gradient rel L2 {rel:.2e} '
          f'(max / gmax {mx:.2e}), pred max-abs {perr:.2e}', flush=True)
    assert perr < 1e-4, perr
    assert rel < 5e-5 and mx < 2.5e-4, (rel, mx)


@pytest.mark.parametrize('case', [(1, 64, 64), (1, 30, 50), (3, 128, 256)])
def test_reflect_pad_adjoint_fused_matches_separate_launches(case):
    """lama_reflect_pad_bwd_fused (v108) against fold + add + act_bwd as separate launches, on channel views of the 512-channel state
    (same terms, another order of the additions)."""
    lib = F._DEFAULT_EXEC.lib
    pad, H, W = case
    gen = torch.Generator().manual_seed(11)
    gp = torch.randn(2, 384, H + 2 * pad, W + 2 * pad, generator=gen).to(DEV)
    add1 = torch.randn(2, 384, H, W, generator=gen).to(DEV)
    ident, y = torch.randn(2, 512, H, W, generator=gen).to(DEV), torch.randn(2, 512, H, W, generator=gen).to(DEV)
    fold = torch.empty(2, 384, H, W, device=DEV)
    lib.reflect_pad_bwd(L.view(gp), L.view(add1), pad, L.view(fold), 2)
    s = torch.empty_like(fold)
    lib.add(L.view(fold), L.view(ident, 128, 384), L.view(s), 2)
    sm = torch.empty_like(s)
    lib.act_bwd(L.view(s), L.view(y, 128, 384), L.ACT_RELU, L.view(sm), 2)
    g, gm = ident.clone(), torch.zeros(2, 512, H, W, device=DEV)
    lib.reflect_pad_bwd_fused(L.view(gp), L.view(add1), L.view(g, 128, 384), pad, L.view(y, 128, 384), L.ACT_RELU, L.view(g, 128, 384),
                              L.view(gm, 128, 384), 2)
    torch.cuda.synchronize()
    assert torch.allclose(g[:, 128:], s, atol=1e-5, rtol=0) and torch.allclose(gm[:, 128:], sm, atol=1e-5, rtol=0)      # (the order of the additions differs)
    assert torch.equal(g[:, :128], ident[:, :128]) and float(gm[:, :128].abs().max()) == 0.0
