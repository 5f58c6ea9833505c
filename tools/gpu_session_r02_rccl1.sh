#!/bin/bash
# the RCCL path of bench.py with ONE rank (all a single-GPU box can host): self-spawn through torch.distributed.run, process group on
# 'nccl', all_gather_into_tensor of the u8 outputs, barrier, MAX all-reduce of the time
O=gpurun_out/r02rccl1
mkdir -p $O
LAMA_BENCH_FORCE_DIST=1 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline --no-f32-leg --no-eager-leg > $O/bench_forced_dist.json 2> $O/bench_forced_dist.err
echo "rc=$?"; tail -c 600 $O/bench_forced_dist.err; tail -1 $O/bench_forced_dist.json | cut -c1-400
