#!/bin/bash
# control plane: RCCL forced with two ranks on ONE device (it refuses) -> gloo fallback; and RCCL with one rank (FORCE_DIST)
set -u
mkdir -p gpurun_out/r03_run28
true
true
CL_BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-streaming > gpurun_out/r03_run28/bench_one_rank_rccl.json 2> gpurun_out/r03_run28/bench_one_rank_rccl.err; echo "rc=$?"
python -c "
import json; d=json.load(open('gpurun_out/r03_run28/bench_one_rank_rccl.json')); print({k:d.get(k) for k in ('n_gpus','ranks','world_size_seen','control_backend','control_fallback','ms_per_step')})"
echo "stdout lines: $(wc -l < gpurun_out/r03_run28/bench_one_rank_rccl.json)"
# the same under torch.distributed.run (what the driver does for N > 1), one rank
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-streaming > gpurun_out/r03_run28/bench_torchrun_one_rank.json 2> gpurun_out/r03_run28/bench_torchrun_one_rank.err; echo "rc=$?"
echo "stdout lines: $(wc -l < gpurun_out/r03_run28/bench_torchrun_one_rank.json)"
python -c "
import json; d=json.loads(open('gpurun_out/r03_run28/bench_torchrun_one_rank.json').read().strip().split(chr(10))[0]); print({k:d.get(k) for k in ('n_gpus','ranks','world_size_seen','control_backend','control_fallback','ms_per_step')})"
