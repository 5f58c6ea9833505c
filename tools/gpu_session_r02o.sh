#!/bin/bash
# round 2, session o: is the MFMA ceiling a power / clock ceiling?  sustained MFMA micro-benchmark, zero-operand probe, clocks under the bench
mkdir -p gpurun_out/r02o
O=gpurun_out/r02o
hipcc --offload-arch=gfx950 -O3 -w tools/ubench/mfma_power.hip -o /tmp/mfma_power && /tmp/mfma_power > $O/mfma_power.txt 2>&1
cat $O/mfma_power.txt
export LAMA_HIP_LIB=$PWD/lama_amd/lib/liblama_hip_prof.so
for z in 0 1 0 1; do echo -n "KPROBE_ZERO=$z " >> $O/zero.txt; KPROBE_ZERO=$z KPROBE_ITERS=200 python tools/kprobe.py f16x3 convA convB 2>&1 | grep conv | tr '\n' ' ' >> $O/zero.txt; echo >> $O/zero.txt; done
cat $O/zero.txt
unset LAMA_HIP_LIB
(python bench.py --steps 400 --no-f32-leg --no-cpu-baseline --no-eager-leg > $O/bench_long.json 2>/dev/null) &
BP=$!
sleep 25
for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|mclk|fclk" >> $O/smi.txt; echo "--" >> $O/smi.txt; sleep 0.7; done
wait $BP
cat $O/smi.txt | head -40
python -c "
import json; d=json.loads(open('$O/bench_long.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
