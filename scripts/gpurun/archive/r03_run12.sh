#!/bin/bash
set -u
OUT=gpurun_out/r03_run12
mkdir -p $OUT
# the driver's multi-GPU command form, as far as one GPU allows: (a) one rank under torch.distributed.run with the RCCL control plane forced on,
# (b) two ranks under torch.distributed.run sharing device 0 (gloo control plane), (c) the plain command with self-spawned ranks
CL_BENCH_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 1 --steps 20 --warmup 5 --no-streaming --no-cpu-baseline > $OUT/torchrun_1rank_rccl.json 2>$OUT/torchrun_1rank_rccl.err; echo "rc=$?"
CL_BENCH_OVERSUBSCRIBE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29532 bench.py --gpus 2 --steps 20 --warmup 5 --no-streaming > $OUT/torchrun_2ranks_oversub.json 2>$OUT/torchrun_2ranks_oversub.err; echo "rc=$?"
CL_BENCH_OVERSUBSCRIBE=1 timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 --no-streaming --config C5 > $OUT/self_spawn_2ranks_C5.json 2>$OUT/self_spawn_2ranks_C5.err; echo "rc=$?"
for f in torchrun_1rank_rccl torchrun_2ranks_oversub self_spawn_2ranks_C5; do echo "== $f"; head -c 700 $OUT/$f.json; echo; tail -3 $OUT/$f.err; done
