#!/bin/bash
# round 2, session s: full GPU suite (incl. evaluator / export tests), smoke, bench
mkdir -p gpurun_out/r02s
O=gpurun_out/r02s
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -6 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02s/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('avg_us'), json.dumps(d.get('configs2_fp16_leg'))[:200])
PY
