#!/bin/bash
# eight ranks on the one GPU (oversubscription hook, gloo): self-spawned and under torch.distributed.run -- plumbing at the driver's N
set -u
mkdir -p gpurun_out/r03_run40
( time CL_BENCH_OVERSUBSCRIBE=1 timeout 600 python bench.py --gpus 8 --steps 20 --warmup 5 > gpurun_out/r03_run40/bench_gpus8_selfspawn.json 2> gpurun_out/r03_run40/selfspawn.err ) 2>&1 | grep real; echo "rc=$? lines=$(wc -l < gpurun_out/r03_run40/bench_gpus8_selfspawn.json)"
( time CL_BENCH_OVERSUBSCRIBE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29733 bench.py --gpus 8 --steps 20 --warmup 5 > gpurun_out/r03_run40/bench_gpus8_torchrun.json 2> gpurun_out/r03_run40/torchrun.err ) 2>&1 | grep real; echo "lines=$(wc -l < gpurun_out/r03_run40/bench_gpus8_torchrun.json)"
python - <<'PY'
import json
for f in ('selfspawn','torchrun'):
    txt=open(f'gpurun_out/r03_run40/bench_gpus8_{f}.json').read().strip().splitlines()
    d=json.loads(txt[-1]); print(f, len(txt), {k:d.get(k) for k in ('n_gpus','ranks','world_size_seen','control_backend','oversubscribed','ms_per_step')}, len(d['rank_ms_per_step']), 'hbm_streaming' in d['roofline'])
PY
