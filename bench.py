#!/usr/bin/env python3
"""Benchmark of the LaMa FFC hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): inpainted images/s at 512x512, big-lama, fp32.  One *step* = one pass of the
hot path over one batch of synthetic input per rank: mask compose -> FFCResNetGenerator (big-lama,
random-init weights of that architecture) -> blend -> u8 quantisation, on a batch of 8 images of
512x512 that is already resident in HBM when the timed region starts (BASELINE configs[1];
N ranks = configs[3] per rank).  N > 1: one process per GPU (torch.distributed over RCCL), images are
sharded data-parallel (weak scaling, 8 per rank) and the only data-path collective is the gather
(to the writer rank 0) of the u8 output images, inside the timed region.

The JSON line also carries
  roofline      -- the dominant kernel of the step (by total time): algorithmic FLOPs per launch /
                   its average launch duration, measured here with HIP events on the launch stream
                   in instrumented eager steps after the timed region (graph replay cannot be
                   bracketed per kernel); profiles/ holds the rocprofv3 --kernel-trace --stats summary
                   of this same command for cross-checking.  `peak` is the spec figure (2500 TF dense f16 / 3 products);
                   `peak_sustained` is what THIS box sustains on random operands with nothing but MFMAs in flight (measured
                   live through the profiling build's lama_debug_mfma_peak: the part is power limited) and
                   `frac_of_sustained` prices the kernel against that.  `traffic` comes from the committed PMC passes
                   (profiles/r06_pmc.json: in-pipeline counter passes of this round's launch sequence; a live bench run cannot host the profiler).
  roofline_ffc  -- the unit BASELINE.json names: FourierUnit forward (rfft2 -> spectral 1x1+BN+ReLU ->
                   irfft2 + residual), algorithmic bytes / time against the 8 TB/s HBM peak.
  cpu_baseline  -- the oracle (CPU restatement of the reference, same torch-CPU primitives) timed on
                   the host cores on a bounded sample of the same workload (rank 0, N = 1 only), in the
                   three modes of BASELINE.md section 3: batch-1 loop (bin/predict.py mode; = `value`),
                   one batched forward, and OMP_NUM_THREADS=1.
  pytorch_rocm_eager -- BASELINE configs[1]'s comparator: the same generator as PyTorch-ROCm eager ops
                   (MIOpen convs, rocFFT rfftn / irfftn) on the same GPU, timed in a subprocess outside the
                   timed region (the oracle's functional restatement moved to cuda; /root/reference does not
                   exist on the GPU box).
  value_host_fed -- SURVEY.md 8(d)'s "includes H2D/D2H" rate: the same steps fed from pinned host buffers (round 6: u8 HWC image + u8 mask in as they are on disk, u8 out) over
                   PCIe the way lama_amd.predict serves a directory (HostFedStep: upload of batch k+1, compute of batch k and download of
                   batch k-1 as parallel branches of one captured hipGraph per step); beside it the serial form (copies on the compute
                   stream) and round 4's copy-stream pipeline around plain launches.  `value` stays the resident-input rate the bench
                   contract defines (inputs in HBM when the timed region starts); DESIGN.md section 5 quotes both.

  configs2_fp16_leg / configs4_refine_leg -- the other single-GPU configs of BASELINE.json (4 x 1024^2 with fp16 activations beside the
                   fp32-accurate default; refine_predict on one 2048^2 image), rank 0, N = 1 only.  Never `value`.

`--gpus N` without a torch.distributed environment re-executes itself under torch.distributed.run with N
ranks (one per GPU, RCCL) and relays rank 0's JSON line.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from lama_amd import _lib as L  # noqa: E402
from lama_amd import trainers  # noqa: E402
from lama_amd.predict import gather_to_root  # noqa: E402

BIG_LAMA = dict(
    kind='ffc_resnet', input_nc=4, output_nc=3, ngf=64, n_downsampling=3, n_blocks=18, add_out_act='sigmoid',
    init_conv_kwargs=dict(ratio_gin=0, ratio_gout=0, enable_lfu=False),
    downsample_conv_kwargs=dict(ratio_gin=0, ratio_gout=0, enable_lfu=False),
    resnet_conv_kwargs=dict(ratio_gin=0.75, ratio_gout=0.75, enable_lfu=False))
BATCH, RES = 8, 512            # BASELINE configs[1]; LAMA_BENCH_BATCH / LAMA_BENCH_RES override them for exploratory runs only
BATCH = int(os.environ.get('LAMA_BENCH_BATCH', BATCH))
RES = int(os.environ.get('LAMA_BENCH_RES', RES))
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TF = 157.3       # v_mfma_f32_32x32x2_f32 dense peak
MFMA_BF16_PEAK_TF = 2500.0


def synthetic_batch(device, seed, batch=None, res=None):
    """image uniform [0,1) quantised to u8/255; mask = centred rectangle (25 % of the area) + 3 random strokes."""
    b, r = batch or BATCH, res or RES
    g = torch.Generator().manual_seed(seed)
    img = torch.floor(torch.rand(b, 3, r, r, generator=g) * 256).clamp_(0, 255) / 255.0
    mask = torch.zeros(b, 1, r, r)
    mask[:, :, r // 4: r // 4 + r // 2, r // 4: r // 4 + r // 2] = 1.0
    gs = torch.Generator().manual_seed(4321 + seed)
    for bi in range(b):
        for _ in range(3):
            y0, x0 = int(torch.randint(0, r, (1,), generator=gs)), int(torch.randint(0, r, (1,), generator=gs))
            ln = int(torch.randint(max(2, r // 8), max(3, r // 2), (1,), generator=gs))
            th = max(1, r // 32)
            if int(torch.randint(0, 2, (1,), generator=gs)):
                mask[bi, 0, y0:y0 + th, x0:x0 + ln] = 1.0
            else:
                mask[bi, 0, y0:y0 + ln, x0:x0 + th] = 1.0
    return img.to(device), mask.to(device)


def build_model(device, precision, to_device=True):
    torch.manual_seed(0)
    model = trainers.DefaultInpaintingTrainingModule(dict(generator=dict(BIG_LAMA)))
    # random-init weights; give the BatchNorms non-trivial statistics so nothing folds to identity
    g = torch.Generator().manual_seed(1)
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.weight.data = torch.rand(m.weight.shape, generator=g) + 0.5
            m.bias.data = torch.randn(m.bias.shape, generator=g) * 0.2
            m.running_mean.data = torch.randn(m.bias.shape, generator=g) * 0.1
            m.running_var.data = torch.rand(m.bias.shape, generator=g) + 0.5
    model.freeze()
    if to_device:
        model.to(device)
        model.generator.set_precision(precision)
    return model


class KernelTimer:
    """HIP-event pairs around selected C-ABI launches (eager steps only)."""

    def __init__(self, lib):
        self.lib, self.records, self.on = lib, {}, False
        self.flops, self.bytes, self._nb = {}, {}, {}
        self._conv, self._fu, self._wino = lib.conv2d, lib.fourier_unit, lib.winograd_conv3x3
        lib.conv2d, lib.fourier_unit, lib.winograd_conv3x3 = self.conv2d, self.fourier_unit, self.winograd_conv3x3

    def _timed(self, key, fn, *a, **kw):
        if not self.on:
            return fn(*a, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn(*a, **kw)
        e1.record()
        self.records.setdefault(key, []).append((e0, e1))
        return r

    def conv2d(self, x, w_packed, y, batch, k, *a, **kw):
        x2 = kw.get("x2", a[7] if len(a) > 7 else None)
        tr = kw.get("transposed", a[3] if len(a) > 3 else False)
        key = f'conv{k}x{k}{"T" if tr else ""}_cin{x.C}_cout{y.C}_{y.H}x{y.W}' + (f'+1x1_cin{x2.C}' if x2 is not None else '')
        # algorithmic FLOPs of the launch (2*M*N*K; a transposed conv touches 9/4 taps per output pixel)
        kterm = x.C * k * k / (4.0 if tr else 1.0) + (x2.C if x2 is not None else 0)
        flops = 2.0 * batch * y.H * y.W * y.C * kterm
        # algorithmic bytes of the launch: every operand once (x, the 1x1 segment's x2, the residual, the packed weights), y once
        resid = kw.get('resid', a[6] if len(a) > 6 else None)
        nbytes = 4.0 * batch * (x.C * x.H * x.W + y.C * y.H * y.W + (x2.C * x2.H * x2.W if x2 is not None else 0)
                                + (y.C * y.H * y.W if resid is not None else 0))
        for wp in (w_packed, kw.get('w2_packed')):
            if torch.is_tensor(wp):
                nbytes += wp.numel() * wp.element_size()
        f1 = kw.get('fuse1')
        if f1 is not None:      # SpectralTransform.conv1 of the next layer rides in this launch's epilogue: its flops and its output count here
            key += f'+next_conv1x1_cout{f1[2].C}'
            flops += 2.0 * batch * y.H * y.W * f1[2].C * y.C
            nbytes += 4.0 * batch * f1[2].C * y.H * y.W
        # (the same key is launched with and without a residual operand -- first / second layer of a block: the bytes are the mean over the launches)
        n, mean = self._nb.get(key, (0, 0.0))
        self._nb[key] = (n + 1, (mean * n + nbytes) / (n + 1))
        self.flops[key], self.bytes[key] = flops, self._nb[key][1]
        return self._timed(key, self._conv, x, w_packed, y, batch, k, *a, **kw)

    def fourier_unit(self, x, *a, **kw):
        return self._timed(f'fourier_unit_c{x.C}_{x.H}x{x.W}', self._fu, x, *a, **kw)

    def winograd_conv3x3(self, x, w_packed, y, batch, *a, **kw):
        # the SAME function as the direct 3x3 launch (same key, so every consumer of the table finds the layer); the algorithmic FLOPs are
        # those of the direct convolution: what Winograd removes is MFMA products, not work the layer specifies.  Two launches (GEMM in
        # the transform domain + inverse transform / epilogue), timed together.
        key = f'conv3x3_cin{x.C}_cout{y.C}_{y.H}x{y.W}'
        self.flops[key] = 2.0 * batch * y.H * y.W * y.C * x.C * 9
        self.bytes[key] = 4.0 * batch * (x.C * x.H * x.W + y.C * y.H * y.W)
        self.winograd_keys = getattr(self, 'winograd_keys', set()) | {key}
        return self._timed(key, self._wino, x, w_packed, y, batch, *a, **kw)

    def summary(self):
        out = {}
        for k, evs in self.records.items():
            ts = [a.elapsed_time(b) * 1e3 for a, b in evs]
            out[k] = dict(n=len(ts), avg_us=sum(ts) / len(ts), total_us=sum(ts))
        return out


def pmc_traffic(kernel_key, precision):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes (profiles/*pmc*.json, collected
    with this same workload; a live bench run cannot host the profiler) or None."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, 'profiles', '*pmc*.json')), reverse=True):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        e = d.get(precision, {}).get(kernel_key)
        if e:
            return dict(e, source=os.path.relpath(f, ROOT))
        e = d.get(precision, {}).get(kernel_key.split('+next_')[0])
        if e:       # only the launch WITHOUT the fused conv1 of the next layer was profiled: not the launch that is timed -> no `traffic_bytes`
            e = dict(e, source=os.path.relpath(f, ROOT))
            return dict(traffic_bytes=None, unfused_launch=e,
                        note='the committed counters describe this launch without the fused conv1 epilogue (+25 MB of x1 written); '
                             'they are NOT the traffic of the launch avg_us / flops_per_launch / algorithmic_bytes describe')
    return None


CPU_THREADS_CAP = 32      # oneDNN / MKL stop scaling (and oversubscribe) far below the 128-256 logical cores of the GPU box


def _cpu_state(model=None):
    from oracle import lama_oracle as O     # test infrastructure, used here as the CPU baseline only (never on the GPU path)
    cfg = {k: v for k, v in BIG_LAMA.items() if k != 'kind'}
    if model is None:
        model = build_model('cpu', L.PREC_F32, to_device=False)
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    return O, sd, cfg


def cpu_baseline(model, budget_s=10.0):
    """Oracle (oracle/lama_oracle.py = the reference's arithmetic on torch-CPU, same oneDNN / MKL kernels) on a bounded sample of
    the workload, three modes; `value` is the faithful bin/predict.py mode (batch-1 loop)."""
    import subprocess
    O, sd, cfg = _cpu_state(model)
    ncpu = os.cpu_count() or 1
    nthr = max(1, min(CPU_THREADS_CAP, ncpu))
    torch.set_num_threads(nthr)
    img, mask = synthetic_batch('cpu', 1234)
    with torch.no_grad():
        O.training_module_forward(dict(image=img[:1, :, :128, :128].clone(), mask=mask[:1, :, :128, :128].clone()), sd, cfg)  # warm-up
        n, t_used = 0, 0.0
        while n < BATCH and (n < 2 or t_used < budget_s):
            t0 = time.perf_counter()
            O.training_module_forward(dict(image=img[n:n + 1].clone(), mask=mask[n:n + 1].clone()), sd, cfg)
            t_used += time.perf_counter() - t0
            n += 1
        nb = 4
        t0 = time.perf_counter()
        O.training_module_forward(dict(image=img[:nb].clone(), mask=mask[:nb].clone()), sd, cfg)
        t_b = time.perf_counter() - t0
    modes = {'batch1_loop': dict(images_per_s=round(n / t_used, 4), images=n, threads=nthr, seconds=round(t_used, 1)),
             'batched_forward': dict(images_per_s=round(nb / t_b, 4), images=nb, threads=nthr, seconds=round(t_b, 1))}
    try:    # "as intended by bin/predict.py:16-20": one thread; the variable must be set before torch is imported -> subprocess
        env = dict(os.environ, OMP_NUM_THREADS='1', MKL_NUM_THREADS='1')
        r = subprocess.run([sys.executable, os.path.abspath(__file__), '--cpu-one-thread-leg'], env=env, capture_output=True, text=True, timeout=180)
        modes['omp_num_threads_1'] = json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as e:      # noqa: BLE001
        modes['omp_num_threads_1'] = dict(error=repr(e)[:200])
    return dict(value=round(n / t_used, 4), unit='images/s', cores=nthr, kind='port',
                sample=f'{n} of the {BATCH} 512x512 images, batch-1 loop (bin/predict.py mode), oracle/lama_oracle.py '
                       f'(reference arithmetic on torch-CPU {torch.__version__}), {t_used:.1f} s, torch threads capped at {nthr} of '
                       f'{ncpu} logical cores', modes=modes,
                reference_classes='the reference\'s own FFCResNetGenerator classes cannot travel to the GPU box; timed beside this restatement '
                                  'in the build container (8 cores): bit-identical output, 0.91-1.11x its speed in all three modes '
                                  '(profiles/r02_reference_cpu_timing.txt, tools/ref_cpu_timing.py)')


def cpu_one_thread_leg():
    """Child of cpu_baseline(): OMP_NUM_THREADS=1 was set before torch was imported ("as intended" by bin/predict.py:16-20); one
    512 x 512 image through the oracle."""
    torch.set_num_threads(1)
    O, sd, cfg = _cpu_state(None)
    img, mask = synthetic_batch('cpu', 1234, batch=1)
    with torch.no_grad():
        O.training_module_forward(dict(image=img[:, :, :64, :64].clone(), mask=mask[:, :, :64, :64].clone()), sd, cfg)
        t0 = time.perf_counter()
        O.training_module_forward(dict(image=img.clone(), mask=mask.clone()), sd, cfg)
        dt = time.perf_counter() - t0
    print(json.dumps(dict(images_per_s=round(1.0 / dt, 4), images=1, threads=torch.get_num_threads(), seconds=round(dt, 1))), flush=True)


def configs4_refine_leg(model, device, res=2048, n_iters=15, px_budget=4194304):
    """BASELINE configs[4]: refine_predict on one synthetic res x res image (refiner defaults of configs/prediction/default.yaml with
    px_budget = 4194304 so that the full resolution is kept): seconds per image, second run (plans and packed reverse-pass weights
    cached)."""
    from lama_amd import refinement as R
    g = torch.Generator().manual_seed(5)
    img = torch.rand(1, 3, res, res, generator=g).to(device)
    mask = torch.zeros(1, 1, res, res)
    mask[:, :, res // 4: res // 2, res // 4: 3 * res // 4] = 1.0
    mask = mask.to(device)
    model.generator.use_graph = False
    dts = []
    for _ in range(2):
        batch = dict(image=img, mask=mask, unpad_to_size=[torch.tensor([res]), torch.tensor([res])])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        R.refine_predict(batch, model, gpu_ids='0,', modulo=8, n_iters=n_iters, lr=0.002, min_side=512, max_scales=3, px_budget=px_budget)
        torch.cuda.synchronize()
        dts.append(time.perf_counter() - t0)
    for plan_owner in (model.generator,):
        plan_owner._plans.clear()
    torch.cuda.empty_cache()
    return dict(value=round(dts[-1], 3), unit='s per image', first_run_s=round(dts[0], 3),
                workload=f'big-lama refine_predict, 1 x {res}x{res}, n_iters={n_iters}, px_budget={px_budget}' + (', 3 scales (512 / 1024 / 2048)' if px_budget >= res * res else ' (configs/prediction/default.yaml:24: the image is first rescaled to ~1341^2 -> padded 1344^2, bottleneck planes 168 x 168 = 2^3 3 7, then the pyramid of refinement.py:203-211)') + ', forward f16x3 + reverse pass bf16x3',
                peak_memory_gib=round(torch.cuda.max_memory_allocated() / 2 ** 30, 1))


def photo_leg(model, device, lib, steps=8):
    """VERDICT r5 Next #1: the plane sizes real inputs produce.  Step time of the hot path (u8 in, generator, u8 out; hipGraph replay) at photo-sized
    inputs -- bottleneck planes 135 x 240, 90 x 160, 125 x 188 (= 5^3 x 2^2 47: a large prime factor) -- and at their power-of-two neighbour
    1024 x 2048 on the SAME box; ns per pixel and the ratio to the neighbour's.  Accuracy: the first image of each shape on the parity tests'
    seeded BN-calibrated weights against the CPU fp32 oracle (the checker, outside every timed loop)."""
    shapes = [(1, 1024, 2048), (1, 1080, 1920), (4, 720, 1280), (1, 1000, 1504)]
    out, ref_ns = {}, None
    gen = model.generator
    gen.use_graph = True
    for (b, h, w) in shapes:
        g = torch.Generator().manual_seed(h + w)
        img_u8 = torch.randint(0, 256, (b, h, w, 3), generator=g, dtype=torch.uint8).to(device)
        mask_u8 = torch.zeros(b, h, w, dtype=torch.uint8)
        mask_u8[:, h // 4: h // 4 + h // 2, w // 4: w // 4 + w // 2] = 255
        mask_u8 = mask_u8.to(device)
        u8 = torch.empty(b, h, w, 3, dtype=torch.uint8, device=device)
        for _ in range(3):
            model.forward_u8(img_u8, mask_u8, None, u8)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            model.forward_u8(img_u8, mask_u8, None, u8)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        ns = dt * 1e9 / (b * h * w)
        if ref_ns is None:
            ref_ns = ns
        out[f'{b}x{h}x{w}'] = dict(ms_per_step=round(dt * 1e3, 3), images_per_s=round(b / dt, 2), ns_per_pixel=round(ns, 3),
                                   vs_1024x2048_per_pixel=round(ns / ref_ns, 3), bottleneck_plane=f'{h // 8}x{w // 8}')
        gen._plans.clear()
        torch.cuda.empty_cache()
    try:        # accuracy on calibrated weights (one image per photo shape)
        O, _, cfg = _cpu_state(model)
        torch.set_num_threads(max(1, min(CPU_THREADS_CAP, os.cpu_count() or 1)))
        keep = {k: v.detach().clone() for k, v in model.state_dict().items()}
        sd = {'generator.' + k: v for k, v in O.make_synthetic_state_dict(cfg, seed=0, calib_hw=64).items()}
        model.load_state_dict(sd, strict=True)
        for (b, h, w) in shapes[1:]:
            b1 = O.make_synthetic_batch(1, h, w, seed=h)
            with torch.no_grad():
                ref = O.training_module_forward(dict(image=b1['image'].clone(), mask=b1['mask'].clone()), sd, cfg)['inpainted']
            got = model(dict(image=b1['image'].to(device), mask=b1['mask'].to(device)))['inpainted'].cpu()
            out[f'{b}x{h}x{w}']['max_abs_vs_oracle'] = float(f'{float((got - ref).abs().max()):.3e}')
            gen._plans.clear()
        model.load_state_dict(keep, strict=True)
    except Exception as e:      # noqa: BLE001
        out['accuracy_error'] = repr(e)[:300]
    out['note'] = ('round 6: mixed-radix plane-in-LDS rfft2 / irfft2 for any plane (fft_mr_dev.inc), 128-pixel conv tiles shaped by rounds; the local conv is the '
                   'Winograd form where lama_winograd_preferred says so (any plane size since ABI v110), the direct kernel otherwise.  1 x 1000 x 1504 is 184 tiles of 128 pixels on 256 CUs: '
                   'every launch runs on three quarters of the chip -- its per-pixel ratio is utilisation, not a slow path')
    torch.cuda.empty_cache()
    return out


def predict_cli_leg(model, n_images=512, res=512, io_threads=(8, 16, 32, 64)):
    """VERDICT r5 Next #3: what 8 GPUs will really run -- `python -m lama_amd.predict` end to end on a directory (bin/predict.py:66-94): PNG decode,
    u8 upload, the step, u8 download, PNG encode + write, with plan build / graph capture of the bucket inside the wall time.  512 synthetic
    512 x 512 PNG pairs on tmpfs, a checkpoint directory written from this model; one subprocess per io_threads value.  Also: host CPU-seconds per
    image of the PNG decode (image + mask) and of the encode, single thread, in this process."""
    import shutil
    import subprocess
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    import numpy as np
    import yaml
    from PIL import Image
    root = tempfile.mkdtemp(prefix='lama_cli_', dir='/dev/shm' if os.path.isdir('/dev/shm') else None)
    out = {}
    try:
        mdir, indir = os.path.join(root, 'model'), os.path.join(root, 'in')
        os.makedirs(os.path.join(mdir, 'models'))
        os.makedirs(indir)
        with open(os.path.join(mdir, 'config.yaml'), 'w') as f:
            yaml.safe_dump(dict(training_model=dict(kind='default', concat_mask=True), generator=dict(BIG_LAMA)), f)
        torch.save({'state_dict': {k: v.detach().cpu() for k, v in model.state_dict().items()}}, os.path.join(mdir, 'models', 'best.ckpt'))
        rng = np.random.RandomState(7)
        base = rng.randint(0, 256, (res + 64, res + 64, 3)).astype('uint8')       # uniform noise (the worst case for PNG: incompressible)

        def make(i):
            oy, ox = (i * 7) % 64, (i * 13) % 64
            Image.fromarray(base[oy:oy + res, ox:ox + res]).save(os.path.join(indir, f'im{i:04d}.png'), compress_level=1)
            m = np.zeros((res, res), 'uint8')
            m[res // 4: res // 4 + res // 2, res // 4: res // 4 + res // 2] = 255
            Image.fromarray(m).save(os.path.join(indir, f'im{i:04d}_mask001.png'))

        with ThreadPoolExecutor(min(32, os.cpu_count() or 8)) as ex:
            list(ex.map(make, range(n_images)))
        # host cost of the IO per image (single thread)
        t0 = time.process_time()
        for i in range(32):
            np.asarray(Image.open(os.path.join(indir, f'im{i:04d}.png')).convert('RGB'))
            np.asarray(Image.open(os.path.join(indir, f'im{i:04d}_mask001.png')).convert('L'))
        dec = (time.process_time() - t0) / 32
        arr = np.asarray(Image.open(os.path.join(indir, 'im0000.png')).convert('RGB'))
        t0 = time.process_time()
        from lama_amd.predict import _write_png
        for i in range(32):
            _write_png(os.path.join(root, 'enc.png'), arr)
        enc = (time.process_time() - t0) / 32
        t0 = time.process_time()
        for i in range(8):
            Image.fromarray(arr).save(os.path.join(root, 'enc_pil.png'), compress_level=1)
        enc_pil = (time.process_time() - t0) / 8
        out['host_cpu_s_per_image'] = dict(png_decode_image_and_mask=round(dec, 5), png_encode_result=round(enc, 5), png_encode_result_pil_level1=round(enc_pil, 5),
                                           note='single thread; decode: PIL; encode: lama_amd.predict.encode_png -- filter Sub + zlib level 1 + Z_RLE = cv2.imwrite\'s PNG defaults (what bin/predict.py:94 uses), '
                                                'beside PIL\'s encoder at level 1 (what the CLI used until round 6, third session); uniform-noise images: incompressible, the expensive end of real content')
        runs = {}
        for T in io_threads:
            od = os.path.join(root, f'out{T}')
            t0 = time.perf_counter()
            r = subprocess.run([sys.executable, '-m', 'lama_amd.predict', f'model.path={mdir}', f'indir={indir}', f'outdir={od}', f'io_threads={T}'],
                               cwd=ROOT, capture_output=True, text=True, timeout=600)
            wall = time.perf_counter() - t0
            line = [ln for ln in r.stdout.splitlines() if ln.startswith('wrote ')]
            if r.returncode != 0 or not line:
                runs[str(T)] = dict(error=(r.stderr or r.stdout)[-300:])
                continue
            secs = float(line[-1].split(' in ')[1].split(' s')[0])
            runs[str(T)] = dict(images_per_s=round(n_images / secs, 1), loop_s=round(secs, 3), process_wall_s=round(wall, 1))
            shutil.rmtree(od, ignore_errors=True)
        out['by_io_threads'] = runs
        ok = {int(k): v['images_per_s'] for k, v in runs.items() if 'images_per_s' in v}
        if ok:
            best = max(ok.values())
            sat = min(k for k, v in ok.items() if v >= 0.97 * best)
            out['value'], out['unit'], out['saturates_at_io_threads'] = best, 'images/s', sat
            # the loop time of 512 images is mostly the bucket's set-up (plan build, split-plan timing check, graph capture): the STEADY rate is the slope
            # between this directory and one three times its size (hard links to the same files), at the thread count that saturated
            try:
                big = os.path.join(root, 'in3')
                os.makedirs(big)
                for rep in range(3):
                    for i in range(n_images):
                        os.link(os.path.join(indir, f'im{i:04d}.png'), os.path.join(big, f'r{rep}im{i:04d}.png'))
                        os.link(os.path.join(indir, f'im{i:04d}_mask001.png'), os.path.join(big, f'r{rep}im{i:04d}_mask001.png'))
                od = os.path.join(root, 'out3')
                r = subprocess.run([sys.executable, '-m', 'lama_amd.predict', f'model.path={mdir}', f'indir={big}', f'outdir={od}', f'io_threads={sat}'],
                                   cwd=ROOT, capture_output=True, text=True, timeout=900)
                line = [ln for ln in r.stdout.splitlines() if ln.startswith('wrote ')]
                if r.returncode == 0 and line:
                    secs3 = float(line[-1].split(' in ')[1].split(' s')[0])
                    secs1 = runs[str(sat)]['loop_s']
                    out['steady_state'] = dict(images_per_s=round(2 * n_images / max(secs3 - secs1, 1e-9), 1), loop_s_3x=round(secs3, 3), loop_s_1x=secs1, io_threads=sat,
                                               bucket_setup_s=round(secs1 - n_images * (secs3 - secs1) / (2 * n_images), 3),
                                               note=f'slope between {n_images} and {3 * n_images} images: the rate a long directory sees; bucket_setup_s = the fixed part of the loop time')
                else:
                    out['steady_state'] = dict(error=(r.stderr or r.stdout)[-300:])
                shutil.rmtree(od, ignore_errors=True)
            except Exception as e:      # noqa: BLE001
                out['steady_state'] = dict(error=repr(e)[:300])
            cores = 830.0 * (dec + enc)
            out['host_cores_for_8_ranks_at_830_images_per_s'] = round(8 * cores, 1)
            out['statement'] = (f'one rank at 830 images/s needs {cores:.1f} host cores of PNG decode + encode ({dec * 1e3:.1f} + {enc * 1e3:.1f} ms per image on this host); '
                                f'eight ranks need {8 * cores:.0f} -- of the {os.cpu_count()} logical cores of this box')
        out['workload'] = (f'python -m lama_amd.predict on {n_images} synthetic {res}x{res} PNG pairs on tmpfs (bin/predict.py:66-94 end to end; loop_s includes the plan build + '
                           'graph capture of the one shape bucket, excludes process start and checkpoint load)')
    finally:
        shutil.rmtree(root, ignore_errors=True)
    return out


def sustained_mfma_peak(device, seconds=0.25):
    """Dense f16 MFMA rate THIS box sustains on random operands (nothing but v_mfma_f32_32x32x16_f16 on register operands that change
    with every instruction: lama_debug_mfma_peak in the profiling build of the library, csrc/debug_probes.hip).  The part is power
    limited: all-zero data reach ~98 % of the 2.5 PF spec, random data 67-71 % -- measured here, live."""
    import ctypes as C
    path = os.path.join(ROOT, 'lama_amd', 'lib', 'liblama_hip_prof.so')
    if not os.path.exists(path):
        return None
    try:
        lib = C.CDLL(path)
        fn = lib.lama_debug_mfma_peak
        fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
        g = torch.Generator().manual_seed(7)
        data = (torch.randn(4096 * 8, generator=g) * 1.5).half().to(device)
        st = torch.cuda.current_stream().cuda_stream
        if fn(st, 2000, data.data_ptr(), None) != 0:
            return None
        torch.cuda.synchronize()
        iters = 200000
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn(st, iters, data.data_ptr(), None)
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b)
        tf = 256 * 4 * iters * 16 * 32768.0 / (ms * 1e-3) / 1e12
        return dict(value=round(tf, 1), unit='TFLOP/s dense f16', seconds=round(ms * 1e-3, 3),
                    how='lama_debug_mfma_peak (liblama_hip_prof.so): 256 workgroups x 4 waves, 4 A x 4 B random register fragments per wave, '
                        'v_mfma_f32_32x32x16_f16 only')
    except Exception as e:      # measurement extra: never fail the bench
        return dict(error=str(e))


def eager_leg(steps=5):
    """Child of main(): BASELINE configs[1]'s comparator -- the same generator as PyTorch-ROCm EAGER ops on this GPU (torch conv2d /
    conv_transpose2d = MIOpen, batch_norm, torch.fft.rfftn / irfftn = rocFFT), i.e. the oracle's functional restatement of the
    reference module with its tensors on cuda, fp32, 8 x 512^2, same mask-compose / blend glue, no u8 / gather."""
    O, sd, cfg = _cpu_state(None)
    dev = torch.device('cuda', 0)
    sd = {k: v.to(dev) for k, v in sd.items()}
    img, mask = synthetic_batch(dev, 1234)
    torch.backends.cudnn.benchmark = False
    with torch.no_grad():
        t0 = time.perf_counter()
        O.training_module_forward(dict(image=img.clone(), mask=mask.clone()), sd, cfg)      # MIOpen solver search / kernel JIT
        torch.cuda.synchronize()
        t_first = time.perf_counter() - t0
        O.training_module_forward(dict(image=img.clone(), mask=mask.clone()), sd, cfg)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            O.training_module_forward(dict(image=img.clone(), mask=mask.clone()), sd, cfg)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
    print(json.dumps(dict(value=round(BATCH / dt, 2), unit='images/s', ms_per_step=round(dt * 1e3, 2), steps=steps, dtype='f32',
                          first_call_s=round(t_first, 1), kind='port',
                          note=f'oracle/lama_oracle.py on torch {torch.__version__} ROCm eager (MIOpen + rocFFT), same GPU, outside the timed region')),
          flush=True)


def configs2_leg(model, device, lib, args):
    """4 x 1024^2 (BASELINE configs[2]): PREC_F16 and, for comparison, the default fp32-accurate split at the same shape."""
    out = {}
    img, mask = synthetic_batch(device, 4321, batch=4, res=1024)
    u8 = torch.empty(4, 1024, 1024, 3, dtype=torch.uint8, device=device)
    for name, prec in (('f16', L.PREC_F16), ('f16x3_fp32_activations', L.PREC_F16X3)):
        model.generator.set_precision(prec)
        model.generator.use_graph = not args.no_graph

        def step():
            o = model(dict(image=img, mask=mask))
            lib.quantize_u8_hwc(L.view(o['inpainted']), u8, 4, 1024, 1024, torch.cuda.current_stream().cuda_stream)

        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 8
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        out[name] = dict(value=round(4 / dt, 2), unit='images/s', ms_per_step=round(dt * 1e3, 3), steps=n)
        model.generator._plans.clear()
    out['f16']['speedup_vs_f16x3_path'] = round(out['f16']['value'] / out['f16x3_fp32_activations']['value'], 3)
    # Accuracy of the two paths, measured in this run.  The timed model above has random-init, UNcalibrated weights (its sigmoid output sits at
    # 0.5 +- 0.003, where every path agrees to 5e-5: not an accuracy probe), so one 1024 x 1024 image is run through both paths on the seeded,
    # BN-calibrated weights of the parity tests and compared with the CPU fp32 oracle -- the oracle as the CHECKER, as in smoke(), outside every
    # timed loop.
    try:
        O, _, cfg = _cpu_state(model)
        torch.set_num_threads(max(1, min(CPU_THREADS_CAP, os.cpu_count() or 1)))
        keep = {k: v.detach().clone() for k, v in model.state_dict().items()}
        sd = {'generator.' + k: v for k, v in O.make_synthetic_state_dict(cfg, seed=0, calib_hw=64).items()}
        model.load_state_dict(sd, strict=True)
        b1 = O.make_synthetic_batch(1, 1024, 1024, seed=77)
        with torch.no_grad():
            ref = O.training_module_forward(dict(image=b1['image'].clone(), mask=b1['mask'].clone()), sd, cfg)['inpainted']
        for name, prec in (('f16', L.PREC_F16), ('f16x3_fp32_activations', L.PREC_F16X3)):
            model.generator.set_precision(prec)
            got = model(dict(image=b1['image'].to(device), mask=b1['mask'].to(device)))['inpainted'].cpu()
            d = (got - ref).abs()
            out[name]['max_abs_vs_fp32_oracle'] = float(f'{float(d.max()):.3e}')
            out[name]['mean_abs_vs_fp32_oracle'] = float(f'{float(d.mean()):.3e}')
            model.generator._plans.clear()
        out['accuracy_sample'] = '1 x 1024x1024, seeded BN-calibrated weights (oracle.make_synthetic_state_dict seed 0), CPU fp32 oracle as the checker'
        model.load_state_dict(keep, strict=True)
    except Exception as e:      # noqa: BLE001  (never lose the bench line over a derived figure)
        out['accuracy_error'] = repr(e)[:300]
    out['verdict'] = ('As shipped, LAMA_PREC_F16 is NOT a mode worth choosing: the default path is ~50x closer to the fp32 oracle at ~0.87x the speed '
                      '-- the launches of this network are bound by their request streams, not by the MFMA products fp16 saves (DESIGN.md 4.9); '
                      '<= 5e-3 max-abs is out of reach for any path that spends ONE matrix-core operand per activation (the fp16 read of the '
                      'residual stream alone costs 3.0e-3: profiles/r04_fp16_by_tensor.txt).  It exists because BASELINE configs[2] names it.')
    out['workload'] = 'big-lama 1024x1024 batch=4, mask-compose + generator + blend + u8, hipGraph replay'
    out['dtype_f16'] = ('fp16 activations in HBM from the stem output through the resnet blocks (fp32 residual stream; round 4: the three upsampled tensors and '
                        'the head stay fp32 / 3-term split), weights as hi + lo fp16 parts (2 MFMA products per MAC), fp32 accumulate; the oracle with the '
                        'same storage roundings: 3.4e-3 max / 3.0e-4 mean-abs vs the fp32 oracle at 1x1024^2 (round-3 layout: 1.1e-2; tools/fp16_by_tensor.py)')
    torch.cuda.empty_cache()
    return out


class StepLoop:
    """The timed region's step: mask compose -> generator -> blend -> u8 on a batch that is resident on the device, plus -- with a process
    group -- the only data-path collective: the u8 output images (6.3 MB per rank at 8 x 512^2) gathered to the writer rank 0.  The gather
    runs OFF the compute stream: the u8 batch of step k is gathered (RCCL's own stream, ordered behind a side stream that waits for step k's
    quantize kernel) while step k + 1 computes; two (u8, gathered) buffer pairs, step k + 2 waits for gather k before it overwrites the pair.
    Every gather is inside the timed region: the closing ``barrier()`` waits for the last two.  Device-agnostic (tests/test_dist_gloo.py runs
    it on two gloo ranks with the emulated kernels: no streams / events on a CPU device)."""

    def __init__(self, model, lib, device, img, mask, dist=None, rank=0, world=1):
        self.model, self.lib, self.device, self.img, self.mask = model, lib, torch.device(device), img, mask
        self.dist, self.rank, self.world = dist, rank, world
        self.on_gpu = self.device.type == 'cuda'
        B, _, H, W = img.shape
        self.shape = (B, H, W)
        use_dist = dist is not None
        u8 = torch.empty(B, H, W, 3, dtype=torch.uint8, device=self.device)
        self.u8_ring = [u8, torch.empty_like(u8)] if use_dist else [u8]
        # gather to the writer rank (rank 0), as predict.py does: the other ranks allocate and receive nothing
        self.gathered = [torch.empty(world * B, H, W, 3, dtype=torch.uint8, device=self.device) if rank == 0 else None for _ in range(2)] if use_dist else None
        self.comm_stream = torch.cuda.Stream(device=self.device) if (use_dist and self.on_gpu) else None
        self.quantized = [torch.cuda.Event() for _ in range(2)] if (use_dist and self.on_gpu) else None
        self.gather_work = [None, None]
        self.step_no = 0
        self.gathers = 0
        model.keep_predicted_image = False          # 'inpainted' is all this loop reads (as predict.py): no copy of the generator's output
        # round 6: the step runs the way lama_amd.predict runs it -- u8 HWC image + u8 mask resident in HBM (what is on disk), / 255 and the blend's u8
        # quantisation inside the two elementwise launches around the generator (forward_u8, ABI v110).  The synthetic image IS k / 255 and the mask
        # {0, 1}, so the u8 operands are exact and the result equals the fp32-tensor step's bit for bit (checked once in main()).
        self.img_u8 = (img * 255.0).round().to(torch.uint8).permute(0, 2, 3, 1).contiguous()
        self.mask_u8 = (mask[:, 0] * 255.0).round().to(torch.uint8).contiguous()
        self.use_u8 = hasattr(model, 'forward_u8') and os.environ.get('LAMA_BENCH_U8', '1') != '0'

    def step(self, collect=True):
        use_dist = self.dist is not None and collect
        k = self.step_no & 1 if use_dist else 0
        B, H, W = self.shape
        if use_dist and self.gather_work[k] is not None:
            self.gather_work[k].wait()          # device-side on a GPU: the compute stream waits for gather k - 2 (long done) before reusing its buffers
            self.gather_work[k] = None
        main = torch.cuda.current_stream(self.device) if self.on_gpu else None
        if self.use_u8:
            self.model.forward_u8(self.img_u8, self.mask_u8, None, self.u8_ring[k], binarize=False)
        else:
            out = self.model(dict(image=self.img, mask=self.mask))
            self.lib.quantize_u8_hwc(L.view(out['inpainted']), self.u8_ring[k], B, H, W, main.cuda_stream if self.on_gpu else 0)
        if use_dist:
            if self.on_gpu:
                self.quantized[k].record(main)
                with torch.cuda.stream(self.comm_stream):
                    self.comm_stream.wait_event(self.quantized[k])
                    self.gather_work[k] = gather_to_root(self.dist, self.gathered[k], self.u8_ring[k], self.rank, self.world)
            else:
                self.gather_work[k] = gather_to_root(self.dist, self.gathered[k], self.u8_ring[k], self.rank, self.world)
            self.step_no += 1
            self.gathers += 1

    def barrier(self):
        if self.dist is not None:
            for w_ in self.gather_work:
                if w_ is not None:
                    w_.wait()
            self.dist.barrier()
        if self.on_gpu:
            torch.cuda.synchronize()


def ranks_seen(dist, world, gpus):
    """The line's n_gpus must be the number of ranks that took part in the timed region: process-group size == WORLD_SIZE == --gpus."""
    n_seen = dist.get_world_size() if dist is not None else 1
    assert n_seen == world == gpus, (n_seen, world, gpus)
    return n_seen


def timed_region(loop, steps, warmup):
    """W untimed warm-up steps, then EXACTLY K steps bracketed by a barrier (+ device synchronisation) on both sides; the fp16 split's range flag
    is read ONCE, after the closing barrier and still inside the timed region (no host synchronisation inside a step:
    generator.defer_range_check).  Returns (seconds of this rank -- MAX over the ranks with a process group --, range_ok)."""
    gen = loop.model.generator
    gen.defer_range_check = True
    try:
        for _ in range(warmup):
            loop.step()
        loop.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            loop.step()
        loop.barrier()
        range_ok = gen.check_range(loop.device)
        dt = time.perf_counter() - t0
    finally:
        gen.defer_range_check = False
    if loop.dist is not None:
        tmax = torch.tensor([dt], device=loop.device, dtype=torch.float64)
        loop.dist.all_reduce(tmax, op=loop.dist.ReduceOp.MAX)
        dt = float(tmax.item())
    return dt, range_ok


def spawn_command(gpus, argv, port):
    """The torch.distributed.run command line of `python bench.py --gpus N` (one rank per GPU of ONE node, rendezvous on 127.0.0.1) and
    the environment it needs (dmabuf IPC for RCCL)."""
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={int(gpus)}', '--master-addr', '127.0.0.1',
           '--master-port', str(int(port)), os.path.abspath(__file__), *argv]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    return cmd, env


def spawn_ranks(args, argv):
    """`python bench.py --gpus N` without a torch.distributed environment: re-execute under torch.distributed.run (one rank per GPU)."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        port = so.getsockname()[1]
    cmd, env = spawn_command(args.gpus, argv, port)
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--precision', default=os.environ.get('LAMA_PRECISION', 'f16x3'), choices=['f32', 'bf16x3', 'f16x3'])
    ap.add_argument('--no-f32-leg', action='store_true', help='skip the extra exact-fp32 timing leg')
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--split-batch', type=int, default=0, choices=[0, 1, 2, 4], help='parts of the batch as parallel branches of the hipGraph: 0 = the generator\'s rule, verified by timing once per shape (config.split_check); 1 / 2 / 4 force it')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-eager-leg', action='store_true', help='skip the PyTorch-ROCm eager comparator (subprocess)')
    ap.add_argument('--no-host-fed-leg', action='store_true', help='skip the host-fed legs (profiling runs: only whole-batch launches of the one plan in the trace)')
    ap.add_argument('--no-cli-leg', action='store_true', help='skip the end-to-end python -m lama_amd.predict leg (subprocesses)')
    ap.add_argument('--no-photo-leg', action='store_true', help='skip the photo-sized-input leg')
    ap.add_argument('--lib', default=None, help='A/B runs: another build of liblama_hip.so (e.g. lama_amd/lib/liblama_hip_prof.so, whose kernel '
                                                'switches read LAMA_* variables); the default is the in-tree product library')
    ap.add_argument('--cpu-one-thread-leg', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--eager-leg', action='store_true', help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_one_thread_leg:
        return cpu_one_thread_leg()
    if args.eager_leg:
        return eager_leg()
    if 'WORLD_SIZE' not in os.environ and (args.gpus > 1 or bool(int(os.environ.get('LAMA_BENCH_FORCE_DIST', '0')))):
        raise SystemExit(spawn_ranks(args, sys.argv[1:]))

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    dist = None
    # LAMA_BENCH_FORCE_DIST=1: take the RCCL path (process group, gather to rank 0, barrier, MAX all-reduce) with ONE rank too -- the only way to
    # exercise it on a single-GPU box (launched through torch.distributed.run --nproc-per-node 1)
    use_dist = world > 1 or (bool(int(os.environ.get('LAMA_BENCH_FORCE_DIST', '0'))) and 'RANK' in os.environ)
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        torch.cuda.set_device(local)
        dist.init_process_group('nccl', device_id=torch.device('cuda', local))
    if args.gpus != world:
        # the line's n_gpus must be the number of ranks that actually ran: a mismatch between the flag and the launcher is an error,
        # not something to report around
        raise SystemExit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} (or run '
                         f'`python bench.py --gpus {args.gpus}` without a torch.distributed environment and it spawns the ranks itself)')
    device = torch.device('cuda', local)
    torch.cuda.set_device(device)
    precision = L.PREC_NAMES[args.precision]

    lib = L.use_library(os.path.abspath(args.lib)) if args.lib else L.get_lib()      # raises if the HIP library is missing: no fallback
    timer = KernelTimer(lib)
    model = build_model(device, precision)
    model.generator.use_graph = not args.no_graph
    model.generator.overlap_streams = bool(int(os.environ.get('LAMA_OVERLAP_STREAMS', '1')))   # the generator's default; 0 = serial launch order (A/B runs)
    from lama_amd import ffc as _ffc
    _ffc._DEFAULT_EXEC.local_first = bool(int(os.environ.get('LAMA_LOCAL_FIRST', '1')))
    _ffc._DEFAULT_EXEC.winograd = bool(int(os.environ.get('LAMA_WINOGRAD', '1')))            # the generator's default; 0 for A/B runs
    model.generator.pipeline_local = bool(int(os.environ.get('LAMA_PIPELINE_LOCAL', '0')))   # the generator's default; 1 for A/B runs
    if 'LAMA_FUSE_CONV1' in os.environ:                                                      # default None = by launch order (FFCResNetGenerator._build_plan); 0 / 1 force it (A/B runs)
        model.generator.fuse_conv1 = bool(int(os.environ['LAMA_FUSE_CONV1']))
    model.generator.serial_with_winograd = bool(int(os.environ.get('LAMA_SERIAL_WINOGRAD', '1')))
    img, mask = synthetic_batch(device, 1234 + rank)
    loop = StepLoop(model, lib, device, img, mask, dist=dist if use_dist else None, rank=rank, world=world)
    step, barrier, u8 = loop.step, loop.barrier, loop.u8_ring[0]

    if 'LAMA_INPLACE' in os.environ:          # same-box A/B of the in-place residual state / t over x1 (tools/session.sh ab:LAMA_INPLACE=0,1)
        model.generator.inplace_residual = model.generator.alias_t = bool(int(os.environ['LAMA_INPLACE']))
    if 'LAMA_DEFER_OUT' in os.environ:        # ... of the Winograd output transform inside the next layer's rfft2 launch
        model.generator.defer_wino_out = bool(int(os.environ['LAMA_DEFER_OUT']))
    if 'LAMA_ALIAS_WINO' in os.environ:       # ... of the Winograd partial sums in the FourierUnit's (dead) spectra
        model.generator.alias_wino = bool(int(os.environ['LAMA_ALIAS_WINO']))
    if 'LAMA_SPLIT_BATCH' in os.environ:      # ... of the batch as parallel branches of the graph (0 = the generator's rule, 1 = off, 2 / 4 = forced)
        model.generator.split_batch = int(os.environ['LAMA_SPLIT_BATCH']) or None
    if args.split_batch:                      # --split-batch 1|2|4: override the generator's rule and its timing check
        model.generator.split_batch = args.split_batch
    dt, range_ok = timed_region(loop, args.steps, args.warmup)
    nsplit = model.generator._split_parts((BATCH, 4, RES, RES), device)          # (after the generator's own verification of the split plan)
    split_check = {f'{k[0][0]}x{k[0][2]}x{k[0][3]}': v for k, v in getattr(model.generator, 'split_decisions', {}).items()} or None   # every shape tune_split decided in this process
    if not range_ok:
        raise SystemExit('bench.py: an activation left the fp16 split\'s range during the timed steps: the run is void')

    # the same K steps fed from / drained to pinned HOST buffers (SURVEY.md 8(d) "includes H2D/D2H"): u8 image + mask in, u8 out,
    # copies on the compute stream (serial, not overlapped).  Reported beside `value`, never as `value`.
    dt_pcie = dt_piped = dt_piped_graph = dt_replay = dt_host = hf_mode = None
    if world == 1 and not args.no_host_fed_leg:
        h_img, h_mask = loop.img_u8.cpu().pin_memory(), loop.mask_u8.cpu().pin_memory()      # (round 6) what is on disk: u8 HWC image + u8 mask, 8.4 MB per step
        h_u8 = torch.empty(BATCH, RES, RES, 3, dtype=torch.uint8).pin_memory()
        d_img, d_mask = torch.empty_like(loop.img_u8), torch.empty_like(loop.mask_u8)
        # the u8 step against the fp32-tensor step of round 5 (the reference's host code: / 255, then the module's forward + u8): the same bits
        ref_u8 = torch.empty_like(u8)
        lib.quantize_u8_hwc(L.view(model(dict(image=img, mask=mask))['inpainted']), ref_u8, BATCH, RES, RES, torch.cuda.current_stream().cuda_stream)
        u8_step_bit_identical = bool(torch.equal(ref_u8, model.forward_u8(loop.img_u8, loop.mask_u8, None, torch.empty_like(u8), binarize=False)))
        assert u8_step_bit_identical, 'forward_u8 differs from forward + quantize_u8_hwc'

        def step_pcie():
            d_img.copy_(h_img, non_blocking=True)
            d_mask.copy_(h_mask, non_blocking=True)
            model.forward_u8(d_img, d_mask, None, u8, binarize=False)
            h_u8.copy_(u8, non_blocking=True)

        step_pcie()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step_pcie()
        torch.cuda.synchronize()
        dt_pcie = time.perf_counter() - t1

        # ... and pipelined the way predict.py serves a directory (lama_amd.predict.HostFedStep): double-buffered device inputs / outputs; the H2D of
        # batch k + 1 and the D2H of batch k - 1 beside the compute of batch k -- on copy streams around plain launches (the default), and as
        # parallel branches of ONE captured hipGraph per step (measured slower: the graph executor runs its memcpy nodes in line)
        from lama_amd.predict import HostFedStep

        def host_fed(mode):
            hs = HostFedStep(model, BATCH, RES, RES, device, drain=True, binarize=False, mode=mode)
            for q in range(2):
                hs.h_img[q].copy_(h_img)
                hs.h_mask[q].copy_(h_mask)
                hs.h_sizes[q][:] = torch.tensor([RES, RES], dtype=torch.int32)

            def run(nsteps):
                hs.prime(0)
                for k in range(nsteps):
                    hs.launch(k & 1)
                hs.flush((nsteps - 1) & 1)

            run(3)                                          # warm-up (graph mode: captures both graphs)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            run(args.steps)
            dt2 = time.perf_counter() - t2
            assert torch.equal(hs.h_u8[(args.steps - 1) & 1], h_u8) and (args.steps < 2 or torch.equal(hs.h_u8[args.steps & 1], h_u8))   # the serial leg's images
            return dt2

        dt_replay = host_fed('replay')
        hf_mode = HostFedStep(model, BATCH, RES, RES, device).mode          # what `auto` (predict.py's default) picks for this shape
        dt_piped = host_fed('streams')
        dt_piped_graph = host_fed('graph')
        dt_host = host_fed('host')
        model.generator._plans.clear()

    # instrumented eager steps: per-kernel durations with HIP events on the launch stream
    roof = roof_ffc = None
    kern, kern_seq = {}, {}
    if rank == 0:
        model.generator.use_graph = False
        model.generator.overlap_streams = False     # per-kernel events need every launch on the current stream
        from lama_amd import ffc as _ffc2
        _ffc2._DEFAULT_EXEC.cooperative_serial = True   # ... of the same kernel geometry as the timed region's (FFC.launch sets the flag when it forks)
        # per-UNIT times: in the timed region the Winograd output transform of layer l rides in the rfft2 launch of layer l + 1
        # (generator.defer_wino_out); here every unit runs its own launches, so that `fourier_unit_*` is the FourierUnit alone and
        # `conv3x3_cin512_*` the local conv with both of its launches (the fused launch is in profiles/*kernel_stats*.csv)
        defer_timed = model.generator.defer_wino_out
        # per-KERNEL times need every launch alone on the chip: the parts of the batch that the timed region runs side by side
        # (generator.split_batch) would overlap under the event pairs, so these steps run the one-part plan -- the same kernels over the whole batch
        split_timed = model.generator.split_batch
        model.generator.split_batch = 1
        kern_seq = {}
        if defer_timed:
            # first the timed region's OWN launch sequence (VERDICT r4 Next #4): `fourier_unit_*` then brackets rfft2_ip64_wino_out_kernel (the
            # FourierUnit's first launch WITH the previous layer's Winograd output transform riding in it) + the spectral GEMM + irfft2, and
            # `conv3x3_cin512_*` the Winograd GEMM launch alone -> kernels_us_in_sequence
            model.generator._plans.clear()
            step(collect=False)
            torch.cuda.synchronize()
            timer.on = True
            for _ in range(3):
                step(collect=False)
            torch.cuda.synchronize()
            timer.on = False
            kern_seq = timer.summary()
            timer.records = {}
        model.generator.defer_wino_out = False
        model.generator._plans.clear()
        step(collect=False)                         # rank 0 only: no collective in here (the other ranks are done)
        torch.cuda.synchronize()
        timer.on = True
        for _ in range(3):
            step(collect=False)
        torch.cuda.synchronize()
        timer.on = False
        kern = timer.summary()
        _ffc2._DEFAULT_EXEC.cooperative_serial = False
        model.generator.defer_wino_out = defer_timed
        model.generator.split_batch = split_timed
        model.generator.overlap_streams = bool(int(os.environ.get('LAMA_OVERLAP_STREAMS', '1')))   # back to the timed configuration
        model.generator._plans.clear()
        dom = max((k for k in kern if k.startswith('conv')), key=lambda k: kern[k]['total_us'])
        # the two bottleneck 3x3 launches are within a few per cent of each other in serial order and which one leads flipped from run to
        # run: among the kernels within 5 % of the leader take the one with the most algorithmic FLOPs per step (the local conv -- also the
        # top kernel of the timed region's own rocprofv3 summary, profiles/r02_close_kernel_stats_overlap_on.csv); the other is `runner_up`
        near = [k for k in kern if k.startswith('conv') and timer.flops.get(k) and kern[k]['total_us'] >= 0.95 * kern[dom]['total_us']]
        if len(near) > 1:
            dom = max(near, key=lambda k: timer.flops[k] * kern[k]['n'])
        h = RES // 8
        flops = timer.flops.get(dom)
        peak = MFMA_F32_PEAK_TF if precision == L.PREC_F32 else MFMA_BF16_PEAK_TF / 3.0
        if flops:
            ach = flops / kern[dom]['avg_us'] / 1e6
            roof = dict(kernel=dom, bound='mfma', achieved=round(ach, 2), peak=round(peak, 1), unit='TFLOP/s', frac=round(ach / peak, 4),
                        traffic=(pmc_traffic(dom, args.precision) or {}).get('traffic_bytes'), traffic_detail=pmc_traffic(dom, args.precision),
                        avg_us=round(kern[dom]['avg_us'], 2), flops_per_launch=flops,
                        measured='HIP events around every launch in 3 eager steps of the ONE-PART plan after the timed region, on the launch stream: the launch '
                                 'over the whole batch alone on the chip (the timed region runs the batch as parallel parts -- config.split_batch -- whose '
                                 'quarter-size launches overlap; conv1 of the next layer rides in this launch when the key says +next_conv1x1).  '
                                 'rocprofv3 --kernel-trace --stats of the same command: profiles/r06_kernel_stats.csv',
                        algorithmic_bytes=timer.bytes.get(dom), launches_per_step=kern[dom]['n'] // 3,
                        note='exact-fp32 v_mfma_f32_32x32x2_f32 path' if precision == L.PREC_F32 else
                             f'fp32 accuracy via 3-term {args.precision[:-2]} split on v_mfma_f32_32x32x16_{args.precision[:-2]}: peak = 2500 TF dense / 3 MFMA products per '
                             'algorithmic product; achieved counts algorithmic FLOPs only')
            # the runner-up by total time (the two bottleneck 3x3 launches are within a few per cent of each other: which one is
            # "dominant" flips from box to box)
            others = sorted((k for k in kern if k.startswith('conv') and k != dom and timer.flops.get(k)), key=lambda k: -kern[k]['total_us'])
            if others:
                k2 = others[0]
                a2 = timer.flops[k2] / kern[k2]['avg_us'] / 1e6
                roof['runner_up'] = dict(kernel=k2, achieved=round(a2, 2), frac=round(a2 / peak, 4), avg_us=round(kern[k2]['avg_us'], 2),
                                         launches_per_step=kern[k2]['n'] // 3)
            # the local 3x3 conv of the FFC layers as Winograd F(2x2, 3x3) (two launches, timed together): same algorithmic FLOPs as the direct
            # conv it replaces -- the point of the transform is that 16 / 36 of the MFMA products deliver them
            wk = [k for k in getattr(timer, 'winograd_keys', ()) if k in kern]
            if wk:
                k3 = max(wk, key=lambda k: kern[k]['total_us'])
                a3 = timer.flops[k3] / kern[k3]['avg_us'] / 1e6
                roof['local_conv_winograd'] = dict(
                    kernel=k3 + ' = wino_gemm_kernel + wino_out_kernel (lama_winograd_conv3x3_fwd)', achieved=round(a3, 2), frac=round(a3 / peak, 4),
                    avg_us=round(kern[k3]['avg_us'], 2), launches_per_step=kern[k3]['n'] // 3, flops_per_launch=timer.flops[k3],
                    mfma_products_frac=round(a3 / peak * 16.0 / 36.0, 4),
                    traffic=(pmc_traffic(k3, args.precision) or {}).get('traffic_bytes'), traffic_detail=pmc_traffic(k3, args.precision),
                    note='achieved / frac: algorithmic FLOPs of the 3x3 conv (2*M*N*K*9) over the time of both launches, against the same 833 TF '
                         '3-product ceiling as the direct kernel it replaced (0.48 in round 2); mfma_products_frac: the matrix-core rate actually '
                         'sustained, i.e. 16/36 of that (the transform removes 2.25x of the products)')
            if precision != L.PREC_F32:
                sus = sustained_mfma_peak(device)
                if sus and sus.get('value'):
                    roof['peak_sustained'] = sus
                    roof['frac_of_sustained'] = round(ach / (sus['value'] / 3.0), 4)
        fu = next((k for k in kern if k.startswith('fourier_unit')), None)
        if fu:
            alg = 2 * BATCH * 192 * h * h * 4 + 384 * 384 * 4 + 384 * 4       # read x, write y, weights once
            gbs = alg / kern[fu]['avg_us'] / 1e3
            parts = [pmc_traffic(k, args.precision) for k in ('rfft2_192x64x64', 'conv1x1_cin384_cout384_64x33', 'irfft2_192x64x64')]
            ffc_traffic = sum(p_['traffic_bytes'] for p_ in parts) if all(parts) else None
            # what THIS design can reach at best: three launches that each move their own bytes once (x -> spectrum -> spectrum -> y + the
            # residual re-read) at the measured-achievable 6.3 TB/s copy rate (MI355X_MICROARCH.md), launch boundaries not counted
            three = (parts[0]['algorithmic_bytes'] + parts[1]['algorithmic_bytes'] + parts[2]['algorithmic_bytes']) if all(parts) else \
                int(3.5 * alg)
            ceil_us = three / 6.3e6
            roof_ffc = dict(unit_of_work=f'FourierUnit forward [{BATCH},192,{h},{h}] fp32 (3 launches)', bound='hbm', achieved=round(gbs, 1),
                            peak=HBM_PEAK_GBS, unit='GB/s', frac=round(gbs / HBM_PEAK_GBS, 4), traffic=ffc_traffic,
                            traffic_source='replayed from the committed PMC passes (profiles/*pmc*.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, '
                                           '2 * FETCH + WRITE), NOT measured in this run: a live bench run cannot host the profiler',
                            avg_us=round(kern[fu]['avg_us'], 2), algorithmic_bytes=alg,
                            frac_of_6_29_TBs=round(gbs / 6290.0, 4),
                            in_sequence=None if fu not in kern_seq else dict(
                                avg_us=round(kern_seq[fu]['avg_us'], 2),
                                traffic=(lambda ps: sum(p_['traffic_bytes'] for p_ in ps) if all(ps) else None)(
                                    [pmc_traffic(k, args.precision) for k in ('rfft2_192x64x64+wino_out', 'conv1x1_cin384_cout384_64x33', 'irfft2_192x64x64')]),
                                note='the three launches as the timed region issues them: the first is rfft2_ip64_wino_out_kernel, i.e. rfft2 PLUS the '
                                     'Winograd output transform of the previous layer\'s local conv (67 MB of its own) riding in the FFT workgroups\' '
                                     'HBM-idle transform phase; avg_us above is the FourierUnit alone (plain rfft2 launch)'),
                            ceiling_three_launch=dict(bytes=three, us=round(ceil_us, 2), frac=round(alg / ceil_us / 1e3 / HBM_PEAK_GBS, 4),
                                                      note='the cap of the three-launch design itself: both fp32 spectra round-trip through '
                                                           'memory (181 MB against 51 MB algorithmic) at 6.3 TB/s; fp16-stored spectra / a 2-product GEMM would '
                                                           'lift it but cost 5-7e-4 max-abs end to end (profiles/r05_fu_two_product_accuracy.txt)'))
            try:    # the coarser units beside it (serial-order kernel sums; MFMA utilisation is their primary figure)
                def us(prefix):
                    ks = [k for k in kern if k.startswith(prefix)]
                    return kern[ks[0]]['avg_us'] if ks else None
                t_c1 = us(f'conv1x1_cin384_cout192_{h}x{h}')
                t_loc = us(f'conv3x3_cin512_cout128_{h}x{h}')
                t_glb = us(f'conv3x3_cin128_cout384_{h}x{h}+1x1_cin192')
                if None not in (t_c1, t_loc, t_glb):
                    px = BATCH * h * h
                    st_us = t_c1 + kern[fu]['avg_us']                       # conv2 rides in the global launch (counted there)
                    lay_us = st_us + t_loc + t_glb
                    st_b = 2 * BATCH * 384 * h * h * 4 + (2 * 384 * 192 + 384 * 384) * 4
                    lay_b = 2 * BATCH * 512 * h * h * 4 + 1329280 * 4
                    lay_f = 2.0 * px * (512 * 128 * 9 + 128 * 384 * 9 + 384 * 192 + 192 * 384) + 2.0 * BATCH * h * (h // 2 + 1) * 384 * 384
                    roof_ffc['other_units'] = {
                        'SpectralTransform (conv1 + FourierUnit; conv2 is a K-segment of the global launch)': dict(
                            us=round(st_us, 1), algorithmic_bytes=st_b, hbm_frac=round(st_b / st_us / 1e3 / HBM_PEAK_GBS, 4)),
                        'FFC_BN_ACT (5 launches, serial order)': dict(
                            us=round(lay_us, 1), algorithmic_bytes=lay_b, hbm_frac=round(lay_b / lay_us / 1e3 / HBM_PEAK_GBS, 4),
                            gflop=round(lay_f / 1e9, 1), mfma_tflops=round(lay_f / lay_us / 1e6, 1),
                            mfma_frac=round(lay_f / lay_us / 1e6 / (MFMA_F32_PEAK_TF if precision == L.PREC_F32 else MFMA_BF16_PEAK_TF / 3.0), 4)),
                        'FFCResnetBlock (2 layers)': dict(us=round(2 * lay_us, 1), algorithmic_bytes=2 * BATCH * 512 * h * h * 4 + 2 * 1329280 * 4,
                                                         gflop=round(2 * lay_f / 1e9, 1)),
                    }
            except Exception as e:      # noqa: BLE001  (never lose the bench line over a derived figure)
                roof_ffc['other_units'] = dict(error=repr(e)[:200])

    # extra leg (rank 0, N = 1): the same step on the exact-fp32 MFMA path, for reference beside the default bf16x3 split
    f32_leg = None
    if rank == 0 and world == 1 and precision != L.PREC_F32 and not args.no_f32_leg:
        model.generator.set_precision(L.PREC_F32)
        model.generator.use_graph = not args.no_graph
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        nf = max(3, args.steps // 4)
        for _ in range(nf):
            step()
        torch.cuda.synchronize()
        d1 = time.perf_counter() - t1
        f32_leg = dict(value=round(BATCH * nf / d1, 3), unit='images/s', ms_per_step=round(d1 / nf * 1e3, 3), steps=nf,
                       note='same workload on v_mfma_f32_32x32x2_f32 (exact fp32, 157.3 TF peak)')
        model.generator.set_precision(precision)

    # extra leg (rank 0, N = 1): BASELINE configs[2] -- big-lama 1024x1024 batch=4 "fp16" (PREC_F16: fp16 activations in HBM, fp16
    # weights, one MFMA product per MAC), beside the same shape on the fp32-accurate default.  A separate config, never the headline.
    c3_leg = None
    if rank == 0 and world == 1 and not args.no_f32_leg and BATCH == 8 and RES == 512:
        try:
            c3_leg = configs2_leg(model, device, lib, args)
        except Exception as e:      # noqa: BLE001
            c3_leg = dict(error=repr(e)[:300])
        model.generator.set_precision(precision)

    # extra leg (rank 0, N = 1): the same network with SIXTEEN images per step (two batches of 8 in flight: four parts of four images as parallel
    # branches of the graph).  Not BASELINE's batch (8 per GPU) and never `value`: it shows what the chip does once the parts oversubscribe it
    b16_leg = None
    if rank == 0 and world == 1 and not args.no_f32_leg and BATCH == 8 and RES == 512:
        try:
            img16, mask16 = torch.cat([img, img.flip(0)], 0), torch.cat([mask, mask.flip(0)], 0)
            loop16 = StepLoop(model, lib, device, img16, mask16)
            model.generator.use_graph = not args.no_graph
            d16, ok16 = timed_region(loop16, max(5, args.steps // 2), 3)
            b16_leg = dict(value=round(16 * max(5, args.steps // 2) / d16, 2), unit='images/s', ms_per_step=round(d16 / max(5, args.steps // 2) * 1e3, 3),
                           range_ok=bool(ok16), split_batch=model.generator._split_parts((16, 4, RES, RES), device),
                           note='16 x 512^2 per step (a serving loop may choose its batch: python -m lama_amd.predict batch_size=16); BASELINE configs[1] is 8')
            del loop16, img16, mask16
            model.generator._plans.clear()
            torch.cuda.empty_cache()
        except Exception as e:      # noqa: BLE001
            b16_leg = dict(error=repr(e)[:300])

    # extra leg (rank 0, N = 1): BASELINE configs[4] -- refinement of one 2048x2048 image (px_budget 4194304, 15 iterations, 3 scales)
    c5_leg = None
    if rank == 0 and world == 1 and not args.no_f32_leg and BATCH == 8 and RES == 512:
        try:
            c5_leg = configs4_refine_leg(model, device)
        except Exception as e:      # noqa: BLE001
            c5_leg = dict(error=repr(e)[:300])
        model.generator.set_precision(precision)
        model.generator.use_graph = not args.no_graph

    # extra legs (rank 0, N = 1), round 6: the default refinement budget, photo-sized inputs, the CLI end to end.  Never `value`.
    c5d_leg = ph_leg = cli_leg = None
    if rank == 0 and world == 1 and not args.no_f32_leg and BATCH == 8 and RES == 512:
        for name, fn in (('c5d', lambda: configs4_refine_leg(model, device, px_budget=1800000)), ('photo', lambda: photo_leg(model, device, lib)),
                         ('cli', lambda: predict_cli_leg(model))):
            if (name == 'cli' and args.no_cli_leg) or (name == 'photo' and args.no_photo_leg):
                continue
            try:
                r_ = fn()
            except Exception as e:      # noqa: BLE001
                r_ = dict(error=repr(e)[:300])
            if name == 'c5d':
                c5d_leg = r_
            elif name == 'photo':
                ph_leg = r_
            else:
                cli_leg = r_
            model.generator.set_precision(precision)
            model.generator.use_graph = not args.no_graph
            model.generator._plans.clear()

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(model)

    eager = None
    if rank == 0 and world == 1 and not args.no_eager_leg and BATCH == 8 and RES == 512:
        import subprocess
        try:
            torch.cuda.empty_cache()
            r = subprocess.run([sys.executable, os.path.abspath(__file__), '--eager-leg'], capture_output=True, text=True, timeout=420)
            eager = json.loads(r.stdout.strip().splitlines()[-1])
        except Exception as e:      # noqa: BLE001
            eager = dict(error=repr(e)[:300])

    if rank == 0:
        n_seen = ranks_seen(dist if use_dist else None, world, args.gpus)
        total_images = world * BATCH * args.steps
        dt_hf = None if dt_replay is None else {'graph': dt_piped_graph, 'replay': dt_replay, 'host': dt_host, 'streams': dt_piped}[hf_mode]
        line = {
            'metric': f'inpainted images/sec at {RES}x{RES} big-lama',
            'value': round(total_images / dt, 3), 'unit': 'images/s', 'n_gpus': world, 'n_ranks_seen': n_seen,
            'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32' if precision == L.PREC_F32 else f'f32 (3-term {args.precision[:-2]} split on the 16-bit MFMA, fp32 accumulate, fp32 activations)', 'data': 'synthetic',
            'config': {'workload': f'big-lama FFCResNetGenerator {RES}x{RES} batch={BATCH}/GPU fp32 (BASELINE configs[1]), '
                                   f'mask-compose + generator + blend + u8, random-init weights',
                       'global_batch': world * BATCH, 'resolution': RES, 'parallelism': f'dp{world}',
                       'hip_graph': not args.no_graph, 'precision': args.precision,
                       'split_batch': f'{nsplit} parts of {BATCH // nsplit} images as parallel branches of the one hipGraph (generator.split_batch)' if nsplit > 1 else 1,
                       'split_check': split_check},
            'roofline': roof, 'roofline_ffc': roof_ffc, 'cpu_baseline': cpu, 'exact_f32_leg': f32_leg,
            'pytorch_rocm_eager': eager, 'configs2_fp16_leg': c3_leg, 'configs4_refine_leg': c5_leg, 'configs4_refine_default_leg': c5d_leg, 'photo_leg': ph_leg, 'predict_cli_leg': cli_leg, 'batch16_leg': b16_leg,
            'value_host_fed': None if dt_replay is None else dict(
                host_synchronised_copy_streams=dict(value=round(BATCH * args.steps / dt_host, 3), ms_per_step=round(dt_host / args.steps * 1e3, 3),
                    note='HostFedStep(mode=host), round 6: the compute queued first, then the HOST waits for the previous step and queues the next upload / previous download on the copy streams -- no device-side fork / join between the queues'),
                u8_step_bit_identical_to_fp32_tensor_step=u8_step_bit_identical,
                value=round(BATCH * args.steps / dt_hf, 3), unit='images/s',
                ms_per_step=round(dt_hf / args.steps * 1e3, 3),
                vs_resident=round(dt / dt_hf, 4), mode=hf_mode,
                note=f'SURVEY.md 8(d) metric (i) "includes H2D/D2H": the same {args.steps} steps fed from pinned host buffers (round 6: u8 HWC image + u8 mask as on disk, '
                     f'{BATCH * 4 * RES * RES / 1e6:.1f} MB in -- / 255, padding and mask > 0 run on the device; u8 images, {BATCH * 3 * RES * RES / 1e6:.1f} MB out per step over PCIe), the way '
                     'lama_amd.predict serves a directory: HostFedStep(mode=auto) -- H2D of batch k+1, compute of batch k, D2H of batch k-1 per launch, in '
                     'the form `mode` names (graph: one captured hipGraph per step with the copies as branches beside the parts of the batch; replay: copy '
                     'streams beside the replay of the generator\'s graph); outputs equal the serial leg bit for bit.  `value` itself is the '
                     'resident-input rate the bench contract asks for (inputs in HBM when the timed region starts).',
                copy_streams_beside_graph_replay=dict(value=round(BATCH * args.steps / dt_replay, 3), ms_per_step=round(dt_replay / args.steps * 1e3, 3)),
                serial=dict(value=round(BATCH * args.steps / dt_pcie, 3), ms_per_step=round(dt_pcie / args.steps * 1e3, 3),
                            note='copies on the compute stream around the generator\'s own graph replay, nothing overlapped'),
                streams_plain_launches=dict(
                    value=round(BATCH * args.steps / dt_piped, 3), ms_per_step=round(dt_piped / args.steps * 1e3, 3),
                    note='HostFedStep(mode=streams): the same copy streams beside ~270 PLAIN launches per batch -- full overlap, but launch-bound on a slow host'),
                graph_with_copy_nodes=dict(
                    value=round(BATCH * args.steps / dt_piped_graph, 3), ms_per_step=round(dt_piped_graph / args.steps * 1e3, 3),
                    note='HostFedStep(mode=graph): the two copies and the compute as parallel branches of ONE captured hipGraph per step (round 5) -- '
                         'in line with the kernels (= serial) when the compute is one kernel chain, overlapped when the batch runs in parallel parts')),
            'kernels_us': {k: round(v['avg_us'], 1) for k, v in sorted(kern.items(), key=lambda kv: -kv[1]['total_us'])},
            'kernels_us_in_sequence': {k: round(v['avg_us'], 1) for k, v in sorted(kern_seq.items(), key=lambda kv: -kv[1]['total_us'])} if rank == 0 and kern_seq else None,
        }
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.barrier()                  # rank 0 is still busy with its roofline section when the others get here
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
