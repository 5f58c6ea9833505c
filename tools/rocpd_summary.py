#!/usr/bin/env python3
"""Dump the per-kernel summary (rocprofv3 --kernel-trace --stats, rocpd sqlite output) as CSV.
usage: rocpd_summary.py <results.db> [out.csv]"""
import csv
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    rows = list(cur.execute('select name, total_calls, total_duration, average, percentage from top_kernels'))
    out = open(sys.argv[2], 'w', newline='') if len(sys.argv) > 2 else sys.stdout
    w = csv.writer(out)
    w.writerow(['kernel', 'calls', 'total_us', 'avg_us', 'pct'])
    for name, calls, tot, avg, pct in rows:
        w.writerow([name, calls, round(tot, 1), round(avg, 3), round(pct, 3)])


if __name__ == '__main__':
    main()
