#!/usr/bin/env python3
"""MFMA utilisation per kernel from rocprofv3 counter passes over the bench (tools/session.sh mfmabench):
    mfma_util.py <SQ_VALU_MFMA_BUSY_CYCLES summary> <GRBM_GUI_ACTIVE summary> [kernel_stats.csv]
SQ_VALU_MFMA_BUSY_CYCLES counts matrix-pipe busy cycles summed over the chip's 1024 SIMDs (32 per v_mfma_f32_32x32x16_{f16,bf16}: MI355X_MICROARCH.md);
GRBM_GUI_ACTIVE the busy clock cycles of the launch summed over the 8 XCDs (each has its own GRBM) -- so
    busy fraction = MFMA_BUSY / (128 SIMDs per XCD x GUI_ACTIVE),
which bench.py's own debug_mfma_peak_kernel (back-to-back MFMAs on every SIMD) calibrates: 0.96.  The third argument is accepted for the launch counts only; an
effective clock from GUI_ACTIVE / wall time is NOT printed -- the counter pass and the timing pass are different runs at different (profiled) clocks."""
import csv
import re
import sys


def load(path):
    out = {}
    for line in open(path):
        m = re.match(r'(.*?)\s+(\S+)\s+avg=\s*([0-9.]+) n=(\d+)', line.rstrip())
        if m:
            out[m.group(1).strip()] = (float(m.group(3)), int(m.group(4)))
    return out


def main():
    busy, act = load(sys.argv[1]), load(sys.argv[2])
    dur = {}
    if len(sys.argv) > 3:
        try:
            for r in csv.DictReader(open(sys.argv[3])):
                dur[r['kernel'][:60]] = float(r['avg_us'])
        except OSError:
            pass
    print(f'{"kernel (grid)":100s} {"MFMA busy":>10s} {"launches":>9s}')
    for k in sorted(busy, key=lambda k: -busy[k][0] * busy[k][1]):
        if k not in act or act[k][0] <= 0 or busy[k][0] <= 0:
            continue
        frac = busy[k][0] / (128.0 * act[k][0])
        print(f'{k[:100]:100s} {frac:10.3f} {busy[k][1]:9d}')


if __name__ == '__main__':
    main()
