#!/usr/bin/env python3
"""Compile one HIP source for gfx950 with -Rpass-analysis=kernel-resource-usage and print a table
(kernel, VGPRs, AGPRs, SGPRs, spills, occupancy).  usage: tools/kres.py lama_amd/csrc/x.hip [filter]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ''
cmd = ['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fno-gpu-rdc', '-I' + os.path.join(ROOT, 'include'),
       '-I' + os.path.join(ROOT, 'lama_amd', 'csrc'), '-c', src, '-o', '/dev/null', '-Rpass-analysis=kernel-resource-usage']
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r'Function Name: (\S+)', line) or re.search(r' Name: (\S+)', line)
    if m:
        cur = dict(name=subprocess.run(['c++filt', m.group(1)], capture_output=True, text=True).stdout.strip())
        rows.append(cur)
        continue
    for key in ('VGPRs', 'AGPRs', 'TotalSGPRs', 'VGPRs Spill', 'SGPRs Spill', 'Occupancy [waves/SIMD]', 'ScratchSize [bytes/lane]'):
        m = re.search(r'\s' + re.escape(key) + r': (\d+)', line)
        if m and cur is not None:
            cur[key] = int(m.group(1))
for r in rows:
    if flt in r['name']:
        print(f"{r['name'][:90]:90s} v={r.get('VGPRs')} a={r.get('AGPRs')} s={r.get('TotalSGPRs')} vsp={r.get('VGPRs Spill')} "
              f"ssp={r.get('SGPRs Spill')} scr={r.get('ScratchSize [bytes/lane]')} occ={r.get('Occupancy [waves/SIMD]')}")
