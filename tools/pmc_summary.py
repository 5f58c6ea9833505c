#!/usr/bin/env python3
"""Average rocprofv3 counter_collection.csv per (kernel, grid size, counter) -- the grid size tells apart launches of one kernel at
different layer shapes.  usage: pmc_summary.py file.csv"""
import collections
import csv
import sys

acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    k = r.get('Kernel_Name', '')[:130]
    g = r.get('Grid_Size') or r.get('Grid_Size_X') or ''
    if g:
        k = f'{k[:118]} g={g}'
    c = r.get('Counter_Name')
    v = float(r.get('Counter_Value', 0))
    a = acc[(k, c)]
    a[0] += v
    a[1] += 1
for (k, c), (s, n) in sorted(acc.items()):
    print(f'{k:130s} {c:28s} avg={s / n:16.1f} n={n}')
