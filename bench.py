#!/usr/bin/env python3
"""Benchmark of the LaMa FFC hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): inpainted images/s at 512x512, big-lama, fp32.  One *step* = one pass of the
hot path over one batch of synthetic input per rank: mask compose -> FFCResNetGenerator (big-lama,
random-init weights of that architecture) -> blend -> u8 quantisation, on a batch of 8 images of
512x512 that is already resident in HBM when the timed region starts (BASELINE configs[1];
N ranks = configs[3] per rank).  N > 1: one process per GPU (torch.distributed over RCCL), images are
sharded data-parallel (weak scaling, 8 per rank) and the only data-path collective is the all-gather
of the u8 output images, inside the timed region.

The JSON line also carries
  roofline      -- the dominant kernel of the step (by total time): algorithmic FLOPs per launch /
                   its average launch duration, measured here with HIP events on the launch stream
                   in instrumented eager steps after the timed region (graph replay cannot be
                   bracketed per kernel); profiles/ holds the rocprofv3 --kernel-trace --stats summary
                   of this same command for cross-checking.
  roofline_ffc  -- the unit BASELINE.json names: FourierUnit forward (rfft2 -> spectral 1x1+BN+ReLU ->
                   irfft2 + residual), algorithmic bytes / time against the 8 TB/s HBM peak.
  cpu_baseline  -- the oracle (CPU restatement of the reference, same torch-CPU primitives) timed on
                   the host cores on a bounded sample of the same workload (rank 0, N = 1 only).
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


def synthetic_batch(device, seed):
    g = torch.Generator().manual_seed(seed)
    img = torch.floor(torch.rand(BATCH, 3, RES, RES, generator=g) * 256) / 255.0
    mask = torch.zeros(BATCH, 1, RES, RES)
    mask[:, :, RES // 4: 3 * RES // 4, RES // 4: 3 * RES // 4] = 1.0
    return img.to(device), mask.to(device)


def build_model(device, precision):
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
    model.freeze().to(device)
    model.generator.set_precision(precision)
    return model


class KernelTimer:
    """HIP-event pairs around selected C-ABI launches (eager steps only)."""

    def __init__(self, lib):
        self.lib, self.records, self.on = lib, {}, False
        self.flops, self.bytes = {}, {}
        self._conv, self._fu = lib.conv2d, lib.fourier_unit
        lib.conv2d, lib.fourier_unit = self.conv2d, self.fourier_unit

    def _timed(self, key, fn, *a, **kw):
        if not self.on:
            return fn(*a, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn(*a, **kw)
        e1.record()
        self.records.setdefault(key, []).append((e0, e1))

    def conv2d(self, x, w_packed, y, batch, k, *a, **kw):
        x2 = kw.get("x2", a[7] if len(a) > 7 else None)
        tr = kw.get("transposed", a[3] if len(a) > 3 else False)
        key = f'conv{k}x{k}{"T" if tr else ""}_cin{x.C}_cout{y.C}_{y.H}x{y.W}' + (f'+1x1_cin{x2.C}' if x2 is not None else '')
        # algorithmic FLOPs of the launch (2*M*N*K; a transposed conv touches 9/4 taps per output pixel)
        kterm = x.C * k * k / (4.0 if tr else 1.0) + (x2.C if x2 is not None else 0)
        self.flops[key] = 2.0 * batch * y.H * y.W * y.C * kterm
        self.bytes[key] = 4.0 * batch * (x.C * x.H * x.W + y.C * y.H * y.W + (x2.C * x2.H * x2.W if x2 is not None else 0))
        return self._timed(key, self._conv, x, w_packed, y, batch, k, *a, **kw)

    def fourier_unit(self, x, *a, **kw):
        return self._timed(f'fourier_unit_c{x.C}_{x.H}x{x.W}', self._fu, x, *a, **kw)

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
    return None


def cpu_baseline(model, budget_s=20.0):
    """Oracle (oracle/lama_oracle.py = the reference's arithmetic on torch-CPU) on a bounded sample:
    batch-1 loop over 512x512 images as bin/predict.py does, default torch threading."""
    from oracle import lama_oracle as O
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    cfg = {k: v for k, v in BIG_LAMA.items() if k != 'kind'}
    img, mask = synthetic_batch('cpu', 1234)
    n, t_used = 0, 0.0
    with torch.no_grad():
        O.training_module_forward(dict(image=img[:1, :, :128, :128].clone(), mask=mask[:1, :, :128, :128].clone()), sd, cfg)  # warm-up
        while n < BATCH and (n < 2 or t_used < budget_s):
            t0 = time.perf_counter()
            O.training_module_forward(dict(image=img[n:n + 1].clone(), mask=mask[n:n + 1].clone()), sd, cfg)
            t_used += time.perf_counter() - t0
            n += 1
    return dict(value=round(n / t_used, 4), unit='images/s', cores=torch.get_num_threads(), kind='port',
                sample=f'{n} of the {BATCH} 512x512 images, batch-1 loop (bin/predict.py mode), oracle/lama_oracle.py '
                       f'(reference arithmetic on torch-CPU {torch.__version__}), {t_used:.1f} s, host has {os.cpu_count()} logical cores')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--precision', default=os.environ.get('LAMA_PRECISION', 'f16x3'), choices=['f32', 'bf16x3', 'f16x3'])
    ap.add_argument('--no-f32-leg', action='store_true', help='skip the extra exact-fp32 timing leg')
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        torch.cuda.set_device(local)
        dist.init_process_group('nccl', device_id=torch.device('cuda', local))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit('launch multi-GPU runs with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...')
    device = torch.device('cuda', local)
    torch.cuda.set_device(device)
    precision = L.PREC_NAMES[args.precision]

    lib = L.get_lib()                      # raises if the HIP library is missing: no fallback
    timer = KernelTimer(lib)
    model = build_model(device, precision)
    model.generator.use_graph = not args.no_graph
    model.generator.overlap_streams = bool(int(os.environ.get('LAMA_OVERLAP_STREAMS', '0')))   # experimental, see lama_amd/ffc.py
    img, mask = synthetic_batch(device, 1234 + rank)
    u8 = torch.empty(BATCH, RES, RES, 3, dtype=torch.uint8, device=device)
    gathered = torch.empty(world * BATCH, RES, RES, 3, dtype=torch.uint8, device=device) if world > 1 else None

    def step():
        out = model(dict(image=img, mask=mask))
        lib.quantize_u8_hwc(L.view(out['inpainted']), u8, BATCH, RES, RES, torch.cuda.current_stream().cuda_stream)
        if world > 1:
            dist.all_gather_into_tensor(gathered, u8)       # the only data-path collective: output images

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # instrumented eager steps: per-kernel durations with HIP events on the launch stream
    roof = roof_ffc = None
    kern = {}
    if rank == 0:
        model.generator.use_graph = False
        model.generator.overlap_streams = False     # per-kernel events need every launch on the current stream
        model.generator._plans = {}
        step()
        torch.cuda.synchronize()
        timer.on = True
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        timer.on = False
        kern = timer.summary()
        dom = max((k for k in kern if k.startswith('conv')), key=lambda k: kern[k]['total_us'])
        h = RES // 8
        flops = timer.flops.get(dom)
        peak = MFMA_F32_PEAK_TF if precision == L.PREC_F32 else MFMA_BF16_PEAK_TF / 3.0
        if flops:
            ach = flops / kern[dom]['avg_us'] / 1e6
            roof = dict(kernel=dom, bound='mfma', achieved=round(ach, 2), peak=round(peak, 1), unit='TFLOP/s', frac=round(ach / peak, 4),
                        traffic=(pmc_traffic(dom, args.precision) or {}).get('traffic_bytes'), traffic_detail=pmc_traffic(dom, args.precision),
                        avg_us=round(kern[dom]['avg_us'], 2), flops_per_launch=flops,
                        algorithmic_bytes=timer.bytes.get(dom), launches_per_step=kern[dom]['n'] // 3,
                        note='exact-fp32 v_mfma_f32_32x32x2_f32 path' if precision == L.PREC_F32 else
                             f'fp32 accuracy via 3-term {args.precision[:-2]} split on v_mfma_f32_32x32x16_{args.precision[:-2]}: peak = 2500 TF dense / 3 MFMA products per '
                             'algorithmic product; achieved counts algorithmic FLOPs only')
        fu = next((k for k in kern if k.startswith('fourier_unit')), None)
        if fu:
            alg = 2 * BATCH * 192 * h * h * 4 + 384 * 384 * 4 + 384 * 4       # read x, write y, weights once
            gbs = alg / kern[fu]['avg_us'] / 1e3
            parts = [pmc_traffic(k, args.precision) for k in ('rfft2_192x64x64', 'conv1x1_cin384_cout384_64x33', 'irfft2_192x64x64')]
            ffc_traffic = sum(p_['traffic_bytes'] for p_ in parts) if all(parts) else None
            roof_ffc = dict(unit_of_work=f'FourierUnit forward [{BATCH},192,{h},{h}] fp32 (3 launches)', bound='hbm', achieved=round(gbs, 1),
                            peak=HBM_PEAK_GBS, unit='GB/s', frac=round(gbs / HBM_PEAK_GBS, 4), traffic=ffc_traffic,
                            avg_us=round(kern[fu]['avg_us'], 2), algorithmic_bytes=alg)

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

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(model)

    if rank == 0:
        total_images = world * BATCH * args.steps
        line = {
            'metric': f'inpainted images/sec at {RES}x{RES} big-lama',
            'value': round(total_images / dt, 3), 'unit': 'images/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32' if precision == L.PREC_F32 else f'f32 (3-term {args.precision[:-2]} split on the 16-bit MFMA, fp32 accumulate, fp32 activations)', 'data': 'synthetic',
            'config': {'workload': f'big-lama FFCResNetGenerator {RES}x{RES} batch={BATCH}/GPU fp32 (BASELINE configs[1]), '
                                   f'mask-compose + generator + blend + u8, random-init weights',
                       'global_batch': world * BATCH, 'resolution': RES, 'parallelism': f'dp{world}',
                       'hip_graph': not args.no_graph, 'precision': args.precision},
            'roofline': roof, 'roofline_ffc': roof_ffc, 'cpu_baseline': cpu, 'exact_f32_leg': f32_leg,
            'kernels_us': {k: round(v['avg_us'], 1) for k, v in sorted(kern.items(), key=lambda kv: -kv[1]['total_us'])},
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
