#!/bin/bash
# round-2 GPU session A: parity under production selection, race probes, bench
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r02a
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > gpurun_out/r02a/pytest.log 2>&1
for v in "4000 32 full nolocal serial" "4000 64 full fftonly gemmonly conv1only"; do
  timeout 600 python tools/race_probe5.py $v >> gpurun_out/r02a/race.log 2>&1
done
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r02a/bench.json 2> gpurun_out/r02a/bench.err
tail -5 gpurun_out/r02a/pytest.log; tail -20 gpurun_out/r02a/race.log; cat gpurun_out/r02a/bench.json | cut -c1-1500
