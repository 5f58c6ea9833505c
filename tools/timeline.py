#!/usr/bin/env python3
"""Kernel timeline of ONE bench step from a rocprofv3 --kernel-trace rocpd database: start offset, duration, queue and short kernel name of
every dispatch between the last two stem launches -- shows which launches of the two streams really overlap (the spectral branch beside the
local 3x3 conv) and where the main stream waits at the join.   usage: timeline.py <results.db> [out.txt] [k]
(k: the step that ENDS at the k-th stem launch from the end, default 1; k = 0: the shortest step of the trace = a graph replay of the timed region)"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'cbns_\w+::', '', name).replace('void ', '')
    m = re.match(r'(\w+)<([^>]*)>', name)
    return (m.group(1) + '<' + m.group(2).replace(' ', '')[:34] + '>') if m else name[:60]


def main():
    db = sqlite3.connect(sys.argv[1])
    out = open(sys.argv[2], 'w') if len(sys.argv) > 2 else sys.stdout
    cur = db.cursor()
    views = [r[0] for r in cur.execute("select name from sqlite_master where type in ('view', 'table')")]
    src = 'kernels' if 'kernels' in views else next(v for v in views if 'kernel_dispatch' in v)
    cols = [r[1] for r in cur.execute(f'pragma table_info({src})')]
    print('# source', src, 'columns', cols, file=out)
    name_c = 'name' if 'name' in cols else next(c for c in cols if 'name' in c)
    q_c = next((c for c in ('queue_id', 'queue', 'stream_id', 'stream') if c in cols), None)
    rows = list(cur.execute(f'select {name_c}, start, end{", " + q_c if q_c else ""} from {src} order by start'))
    stems = [i for i, r in enumerate(rows) if 'stem7' in r[0]]
    if len(stems) < 2:
        print('# fewer than two stem launches', len(rows), file=out)
        return
    k = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    if k == 0:      # the SHORTEST step of the trace: a hipGraph replay of the timed region (eager and host-fed steps are longer)
        j = min(range(len(stems) - 1), key=lambda i: rows[stems[i + 1]][1] - rows[stems[i]][1])
        lo, hi = stems[j], stems[j + 1]
    else:
        lo, hi = stems[-k - 1], stems[-k]
    t0 = rows[lo][1]
    prev_end = {}
    for r in rows[lo:hi]:
        q = r[3] if q_c else 0
        gap = (r[1] - prev_end[q]) / 1e3 if q in prev_end else 0.0
        prev_end[q] = r[2]
        print(f'{(r[1] - t0) / 1e3:10.1f} us  +{(r[2] - r[1]) / 1e3:7.1f}  q{q}  gap {gap:6.1f}  {short(r[0])}', file=out)
    print(f'# step: {(rows[hi][1] - t0) / 1e3:.1f} us, {hi - lo} dispatches', file=out)


if __name__ == '__main__':
    main()
