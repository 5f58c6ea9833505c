#!/usr/bin/env python3
"""Which kernels ran side by side?  From a rocprofv3 --kernel-trace rocpd database of tools/split_trace.py: the two windows of graph replays (between
the marker launches: one-part plan, split plan) -> wall time per replay, sum of kernel durations, time with >= 1 / >= 2 kernels on the GPU, and the
overlapped time by pair of kernel kinds.   usage: overlap_summary.py <results.db> [n_replays=6]"""
import collections
import re
import sqlite3
import sys


def kind(name):
    for pat, k in (('conv_wr_kernel.*12, 1', 'global 12x1'), ('wino_gemm', 'winograd gemm'), ('gemm1x1_wk', 'spectral gemm'), ('rfft2_ip64', 'rfft2 (+wino out)'),
                   ('irfft2_ip64', 'irfft2'), ('conv_wr_kernel', 'downsample'), ('convt2', 'upsample'), ('stem7', 'stem'), ('head7', 'head'), ('gemm1x1_w4', 'conv1'),
                   ('wino_out', 'wino out')):
        if re.search(pat, name):
            return k
    return 'other'


def main():
    db = sqlite3.connect(sys.argv[1])
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    cur = db.cursor()
    views = [r[0] for r in cur.execute("select name from sqlite_master where type in ('view', 'table')")]
    src = 'kernels' if 'kernels' in views else next(v for v in views if 'kernel_dispatch' in v)
    cols = [r[1] for r in cur.execute(f'pragma table_info({src})')]
    name_c = 'name' if 'name' in cols else next(c for c in cols if 'name' in c)
    rows = list(cur.execute(f'select {name_c}, start, end from {src} order by start'))
    lone = [i for i, r in enumerate(rows) if 'spin_kernel' in r[0]]          # torch.cuda._sleep: the window separators of split_trace.py
    print(f'# {len(rows)} dispatches, {len(lone)} marker launches')
    windows = [(lone[j], lone[j + 1]) for j in range(0, len(lone) - 1, 2)][:2]
    for label, (a, b) in zip(('one-part plan', 'split plan'), windows):
        ks = [(r[1], r[2], kind(r[0])) for r in rows[a + 1:b] if 'spin_kernel' not in r[0]]
        if not ks:
            continue
        t0, t1 = min(k[0] for k in ks), max(k[1] for k in ks)
        ev = sorted([(s, 1, kd) for s, e, kd in ks] + [(e, -1, kd) for s, e, kd in ks])
        active = collections.Counter()
        busy1 = busy2 = 0
        pair = collections.Counter()
        last = ev[0][0]
        for t, d, kd in ev:
            dt = t - last
            tot = sum(active.values())
            if tot >= 1:
                busy1 += dt
            if tot >= 2:
                busy2 += dt
                kinds = sorted(k for k, c in active.items() if c > 0)
                for i, ka in enumerate(kinds):
                    if active[ka] >= 2:
                        pair[(ka, ka)] += dt
                    for kb in kinds[i + 1:]:
                        pair[(ka, kb)] += dt
            active[kd] += d
            last = t
        ssum = sum(e - s for s, e, _ in ks)
        print(f'\n== {label}: {len(ks)} kernel launches in {n} replays')
        print(f'wall {(t1 - t0) / n / 1e3:9.1f} us per replay | sum of kernel durations {ssum / n / 1e3:9.1f} us | >= 1 kernel {busy1 / n / 1e3:9.1f} us | '
              f'>= 2 kernels {busy2 / n / 1e3:9.1f} us ({100.0 * busy2 / max(busy1, 1):.0f} %) | mean concurrency {ssum / max(busy1, 1):.2f}')
        per = collections.Counter()
        for s, e, kd in ks:
            per[kd] += e - s
        print('kernel time per replay by kind (us): ' + ', '.join(f'{k} {v / n / 1e3:.0f}' for k, v in per.most_common()))
        if pair:
            print('overlapped time per replay by pair of kinds (us), top 12:')
            for (ka, kb), v in pair.most_common(12):
                print(f'   {ka:18s} || {kb:18s} {v / n / 1e3:9.1f}')


if __name__ == '__main__':
    main()
