#!/bin/bash
# RCCL forced onto two ranks of one device UNDER torch.distributed.run: symmetric refusal -> gloo fallback with its own store
set -u
mkdir -p gpurun_out/r03_run41
CL_BENCH_OVERSUBSCRIBE=1 CL_BENCH_CONTROL=nccl timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29755 bench.py --gpus 2 --steps 20 --warmup 5 --no-streaming > gpurun_out/r03_run41/out.json 2> gpurun_out/r03_run41/err.txt; echo "rc=$? lines=$(wc -l < gpurun_out/r03_run41/out.json)"
python -c "
import json; d=json.loads(open('gpurun_out/r03_run41/out.json').read().strip().splitlines()[-1]); print({k:d.get(k) for k in ('ranks','world_size_seen','control_backend','rank_ms_per_step')}, (d.get('control_fallback') or '')[:80])"
