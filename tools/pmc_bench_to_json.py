#!/usr/bin/env python3
"""profiles/<tag>_pmc.json (what bench.py's `traffic` fields replay) from the IN-PIPELINE counter passes of tools/session.sh pmcbench:
rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (one counter per run, --kernel-trace only) over `python bench.py --steps 4 --warmup 1`, averaged per
(kernel, grid) by tools/pmc_summary.py.  traffic = (2 x FETCH_SIZE + WRITE_SIZE) KiB: on gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes
(MI355X_MICROARCH.md, HBM section).  The launch of every key is the launch bench.py times under that key (the global-branch launch WITH the fused
conv1 of the next layer is its own key).
usage: pmc_bench_to_json.py <out.json> <pmc_bench_FETCH_SIZE.txt> <pmc_bench_WRITE_SIZE.txt>"""
import json
import re
import sys

B, H = 8, 64
MB = lambda *ch: int(4 * B * H * H * sum(ch))        # fp32 [8, ch, 64, 64] tensors
KEYS = [   # (bench key, [kernel-name regex ...] summed, algorithmic bytes, note)
    ('conv3x3_cin128_cout384_64x64+1x1_cin192+next_conv1x1_cout192', [r'conv_wr_kernel_f16x3<9, 2, 1, 4, 12, 1, 0, 2, (false, ){5}2, true>'],
     (18 * MB(128, 192, 384, 384, 192) + 17 * MB(128, 192, 384, 192)) // 35 + 2064384 + 294912,
     'global 3x3 + 1x1 over t (+ residual in the 18 second layers of a block, not in the 17 first ones: the average of the 35 launches of a step) + fused '
     'conv1 of the next layer: x_l, t, (residual) in; x_g, x1 out; weights once'),
    ('conv3x3_cin128_cout384_64x64+1x1_cin192', [r'conv_wr_kernel_f16x3<9, 2, 1, 4, 12, 1, 0, 2, (false, ){5}2, false>'],
     MB(128, 192, 384, 384) + 2064384, 'the same launch without a next layer (the last block): x_l, t, residual in; x_g out'),
    ('conv3x3_cin512_cout128_64x64', [r'::wino_gemm_kernel', r'::wino_out_kernel'], (18 * MB(512, 128, 128) + 18 * MB(512, 128)) // 36 + 2359296,
     'local 3x3 conv as Winograd F(2x2,3x3): both launches (residual in every second layer)'),
    ('conv1x1_cin384_cout384_64x33', [r'gemm1x1_wk_kernel'], int(2 * 4 * B * 384 * 64 * 33 + 589824), 'spectral 1x1 of the FourierUnit'),
    ('conv1x1_cin384_cout192_64x64', [r'gemm1x1_w4_kernel_f16x3<6, 2, false'], MB(384, 192) + 294912, 'SpectralTransform.conv1 as a launch of its own (first residual layer only)'),
    ('rfft2_192x64x64', [r'^void rfft2_ip64_kernel'], int(MB(192) + 4 * B * 384 * 64 * 33), 'rfft2 of 8 x 192 planes of 64 x 64'),
    ('rfft2_192x64x64+wino_out', [r'rfft2_ip64_wino_out_kernel'], int(MB(192) + 4 * B * 384 * 64 * 33) + 2 * MB(128) + (18 * MB(128, 128) + 17 * MB(128)) // 35,
     'the FourierUnit\'s first launch AS THE TIMED REGION ISSUES IT (round 4): rfft2 of 8 x 192 planes + the Winograd output transform of the previous '
     'layer\'s local conv riding in it (partial sums 33.5 MB in, residual in every second layer, y out)'),
    ('irfft2_192x64x64', [r'^void irfft2_ip64_kernel'], int(2 * MB(192) + 4 * B * 384 * 64 * 33), 'irfft2 + the x + fu(x) add'),
    ('conv7x7_cin64_cout3_512x512', [r'head7_ws_kernel'], int(4 * B * 67 * 512 * 512), 'head'),
    ('conv7x7_cin4_cout64_512x512', [r'stem7_ws_kernel'], int(4 * B * 68 * 512 * 512), 'stem'),
]


def parse(path):
    out = {}
    for line in open(path):
        m = re.match(r'^(.*?)\s+(FETCH_SIZE|WRITE_SIZE)\s+avg=\s*([0-9.]+)\s+n=(\d+)', line.rstrip())
        if m:
            out[m.group(1).strip()] = (float(m.group(3)), int(m.group(4)))
    return out


def main():
    out_path, fpath, wpath = sys.argv[1:4]
    fetch, write = parse(fpath), parse(wpath)
    res = {'_how': __doc__.split('usage:')[0].strip(), 'f16x3': {}}
    for key, pats, alg, note in KEYS:
        f = w = 0.0
        names = []
        for pat in pats:
            hit = [k for k in fetch if re.search(pat, k)]
            if not hit:
                continue
            # the launch over the WHOLE batch: of the shapes launched a few times, the one that moves the most bytes (the timed region runs the batch
            # in parts -- generator.split_batch -- whose launches move 1 / 4 of them, persistent kernels even with the same grid; the instrumented
            # steps of bench.py run the whole batch)
            often = [k_ for k_ in hit if fetch[k_][1] >= 3] or hit
            k = max(often, key=lambda k_: fetch[k_][0])
            f += fetch[k][0]
            w += write.get(k, (0.0, 0))[0]
            names.append(k[:110])
        if not names:
            continue
        tb = int((2 * f + w) * 1024)
        res['f16x3'][key] = dict(kernel=' + '.join(names), traffic_bytes=tb, fetch_size_kb_raw=round(f, 1), write_size_kb=round(w, 1), algorithmic_bytes=alg,
                                 traffic_over_algorithmic=round(tb / alg, 3), note=note, source_pass='in-pipeline (session.sh pmcbench)')
        print(f'{key:64s} traffic {tb / 1e6:8.1f} MB  algorithmic {alg / 1e6:8.1f} MB  x{tb / alg:.2f}')
    json.dump(res, open(out_path, 'w'), indent=1)


if __name__ == '__main__':
    main()
