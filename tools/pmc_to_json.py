#!/usr/bin/env python3
"""Assemble profiles/<tag>_pmc.json (what bench.py's `traffic` fields read) from the summaries of tools/pmc_session.sh passes.
usage: pmc_to_json.py [--merge] <out.json> <how-text> <dir-with-p1/p2/p3.summary.txt> [<dir> ...]   (--merge: keep the entries of an existing <out.json> that the given directories do not replace)
Every directory is one pmc_session over tools/kprobe.py; p1 = FETCH_SIZE, p2 = WRITE_SIZE, p3 = SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT."""
import json
import os
import re
import sys

KEYS = [   # (bench key, kernel-name substrings summed into it, note)
    ('conv3x3_cin512_cout128_64x64', ['wino_gemm_kernel', 'wino_out_kernel'], 'local 3x3 conv as Winograd F(2x2,3x3): wino_gemm_kernel + wino_out_kernel (kprobe wino)', 86250000),
    ('conv3x3_cin128_cout384_64x64+1x1_cin192', ['conv_wr_kernel_f16x3<9, 2, 1, 4, 12, 1'], 'global 3x3 conv + fused 1x1 over t + residual (kprobe convB)', 144530000),
    ('conv1x1_cin384_cout192_64x64', ['gemm1x1_w4_kernel_f16x3<6, 2, false'], 'SpectralTransform.conv1 (kprobe conv1; same kernel name as the spectral GEMM: told apart by grid size)', 75500000),
    ('conv1x1_cin384_cout384_64x33', ['gemm1x1_w4_kernel_f16x3<6, 2, false', 'gemm1x1_wk_kernel_f16x3'], 'spectral 1x1 of the FourierUnit (kprobe fuconv)', 52500000),
    ('rfft2_192x64x64', ['void rfft2_ip64_kernel'], 'rfft2 of 8 x 192 planes of 64 x 64', 51200000),
    ('irfft2_192x64x64', ['void irfft2_ip64_kernel'], 'irfft2 + residual', 76400000),
]


def parse(path):
    out = {}
    if not os.path.exists(path):
        return out
    for line in open(path):
        m = re.match(r'^(.*?)\s+(\S+)\s+avg=\s*([0-9.]+)\s+n=(\d+)', line.rstrip())
        if m:
            out.setdefault(m.group(1).strip(), {})[m.group(2)] = float(m.group(3))
    return out


def main():
    merge = '--merge' in sys.argv
    argv = [a for a in sys.argv if a != '--merge']
    out_path, how = argv[1], argv[2]
    res = {'_how': how, 'f16x3': {}}
    if merge and os.path.exists(out_path):
        res = json.load(open(out_path))
        res['_how'] = res.get('_how', '') + ' | ' + how
    for d in argv[3:]:
        tag = os.path.basename(d.rstrip('/'))
        p = [parse(os.path.join(d, f'p{i}.summary.txt')) for i in (1, 2, 3)]
        for key, subs, note, alg in KEYS:
            names = [k for k in p[0] if any(sb in k for sb in subs)]
            if not names:
                continue
            if key.startswith('conv1x1_cin384_cout192') and 'conv1' not in tag:
                continue
            if key.startswith('conv1x1_cin384_cout384') and 'fuconv' not in tag:
                continue
            fetch = sum(p[0][k].get('FETCH_SIZE', 0.0) for k in names)
            write = sum(p[1].get(k, {}).get('WRITE_SIZE', 0.0) for k in names)
            e = dict(kernel=' + '.join(n[:90] for n in names), traffic_bytes=int((2 * fetch + write) * 1024), fetch_size_kb_raw=round(fetch, 1),
                     write_size_kb=round(write, 1), algorithmic_bytes=alg, note=note, source_pass=tag)
            e['traffic_over_algorithmic'] = round(e['traffic_bytes'] / alg, 3)
            busy = [(p[2].get(k, {}).get('SQ_VALU_MFMA_BUSY_CYCLES'), p[2].get(k, {}).get('GRBM_GUI_ACTIVE')) for k in names]
            busy = [b_ for b_ in busy if b_[0] and b_[1]]
            if busy:
                e['mfma_busy'] = round(max(b_[0] / 1024.0 / (b_[1] / 8.0) for b_ in busy), 3)
                if len(names) > 1:
                    e['mfma_busy_note'] = 'of the launch that has MFMA work (wino_gemm_kernel)'
            res['f16x3'][key] = e
    json.dump(res, open(out_path, 'w'), indent=1)
    print(json.dumps({k: (v['traffic_bytes'], v.get('mfma_busy')) for k, v in res['f16x3'].items()}, indent=1))


if __name__ == '__main__':
    main()
