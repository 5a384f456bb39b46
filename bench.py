"""bench.py -- headline benchmark of the MI355X CityLearn step engine.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json `metric`): the 17-building `citylearn_challenge_2022_phase_all` district tables x
65 536 environments per GPU, fp32, one environment step per kernel launch (`cl_step_f32`, mode A: state lives in
HBM, fresh actions every step).  A "step" advances every (env, building) unit by one time step.  The env batch
is sharded across GPUs with no collective on the data path (weak scaling: per-GPU work is fixed).

Prints ONE JSON line (rank 0).  `value` = building-timesteps/s over all GPUs with inputs resident in HBM.
`roofline` prices the step kernel against HBM (algorithmic bytes per launch / measured launch duration);
`cpu_baseline` is the C restatement of the reference arithmetic (oracle/cl_oracle.c, the "port") timed on this
box's host cores on a bounded sample (rank 0, N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / 'tests'))

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
ENVS_PER_GPU = 65536
GRAPH_CHUNK = 100


def cpu_baseline(spec, tables, seconds: float = 12.0) -> dict:
    """Time oracle/cl_oracle.c (double-precision port of the reference arithmetic, OpenMP over envs) on a bounded
    sample of the same workload."""
    from oracle.c_oracle import COracle
    cores = os.cpu_count() or 1
    os.environ.setdefault('OMP_NUM_THREADS', str(cores))
    E = 4096
    ora = COracle(spec, tables, E)
    rng = np.random.RandomState(0)
    acts = [rng.uniform(-1, 1, size=(ora.n_act_cols, E)).astype(np.float32) for _ in range(4)]
    for t in range(3):
        ora.step(acts[t % 4], t)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(10):
            ora.step(acts[n % 4], 1 + n % (ora.T - 2))
            n += 1
    dt = time.perf_counter() - t0
    return {'value': E * ora.B * n / dt, 'unit': 'building-timesteps/s', 'cores': cores, 'kind': 'port',
            'sample': f'oracle/cl_oracle.c (OpenMP, {cores} threads): 17 buildings x {E} envs x {n} steps in {dt:.1f} s'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5000)
    ap.add_argument('--warmup', type=int, default=300)
    ap.add_argument('--envs-per-gpu', type=int, default=ENVS_PER_GPU)
    ap.add_argument('--no-graph', action='store_true', help='launch every step from Python instead of hipGraph replay')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus and world > 1:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    if args.gpus > 1 and world == 1:
        raise SystemExit('for --gpus N > 1 launch with torch.distributed.run (one process per GPU)')
    torch.cuda.set_device(local_rank)
    device = f'cuda:{local_rank}'
    dist = None
    if world > 1 or os.environ.get('CL_BENCH_FORCE_DIST'):       # the env hook exercises the RCCL path on a 1-GPU box (torchrun, 1 rank)
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        # RCCL prints a version banner on STDOUT when the communicator comes up; stdout must carry the one JSON line only
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group('nccl', device_id=torch.device(device))
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)

    from golden_util import golden
    from citylearn_amd.engine import StepEngine

    g = golden('g2022_all')                      # 17-building 2022_phase_all tables (first 720 hours)
    spec = g.spec()
    tables = spec.episode_tables(0)
    E = args.envs_per_gpu
    eng = StepEngine(tables, E, device=device)
    assert eng.lean
    if os.environ.get('CL_TUNE_ENVMAJOR'):        # tuning hooks: kernel variants
        eng.lib.cl_debug_set_envmajor(int(os.environ['CL_TUNE_ENVMAJOR']))
    if os.environ.get('CL_TUNE_VEC'):
        eng.lib.cl_debug_set_vec(int(os.environ['CL_TUNE_VEC']))
    if os.environ.get('CL_TUNE_LEAN'):
        import ctypes
        eng.lib.cl_debug_set_lean.argtypes = [ctypes.c_int, ctypes.c_int]
        eng.lib.cl_debug_set_lean(int(os.environ['CL_TUNE_LEAN']), int(os.environ.get('CL_TUNE_NW', '0')))
    gen = torch.Generator(device=device).manual_seed(1234 + rank)
    acts = [torch.rand((eng.n_act_cols, E), device=device, generator=gen) * 2 - 1 for _ in range(8)]
    T = eng.n_steps - 1                          # an episode of T+1 rows has T transitions

    def run(i0: int, n: int):
        for i in range(i0, i0 + n):
            eng.step(acts[i % 8], i % T)

    use_graph = not args.no_graph
    graphs = {}
    stream = torch.cuda.Stream(device=device)
    with torch.cuda.stream(stream):
        run(0, min(args.warmup, 50))             # first-touch / module load outside any capture
        stream.synchronize()
        if use_graph:
            # chunks of GRAPH_CHUNK consecutive steps, keyed by their offset in the period lcm(8, T)
            def graph_for(i0: int, n: int):
                key = (i0 % (8 * T), n)
                if key not in graphs:
                    gr = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gr, stream=stream):
                        run(i0, n)
                    graphs[key] = gr
                return graphs[key]

        def advance(i0: int, n: int):
            i = i0
            while i < i0 + n:
                c = min(GRAPH_CHUNK, i0 + n - i)
                if use_graph:
                    graph_for(i, c).replay()
                else:
                    run(i, c)
                i += c

        if use_graph:                            # build every graph the timed region will need, untimed
            i = 0
            while i < args.warmup + args.steps:
                c = min(GRAPH_CHUNK, (args.warmup if i < args.warmup else args.warmup + args.steps) - i)
                graph_for(i, c)
                i += c
            eng.reset()
        advance(0, args.warmup)
        stream.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record(stream)
        advance(args.warmup, args.steps)
        ev1.record(stream)
        stream.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        wall = time.perf_counter() - t0
    ev_ms = ev0.elapsed_time(ev1)

    from citylearn_amd.parallel import reduce_max_seconds
    wall_max = reduce_max_seconds(wall, dist, device)          # MAX over ranks
    ev_max = reduce_max_seconds(ev_ms / 1e3, dist, device)
    units_per_step = eng.n_bldg * E
    bytes_per_unit = eng.algorithmic_bytes_per_unit()
    launch_s = ev_max / args.steps
    achieved = units_per_step * bytes_per_unit / launch_s / 1e9

    traffic = None
    pmc = ROOT / 'profiles' / 'r01_bench_pmc_summary.json'
    if pmc.exists() and E == ENVS_PER_GPU:
        # HBM bytes per launch from the rocprofv3 --pmc passes of this same workload (separate FETCH_SIZE / WRITE_SIZE
        # runs, KiB units; FETCH_SIZE doubled per the gfx950 wide-load correction of MI355X_MICROARCH.md)
        c = json.loads(pmc.read_text())
        traffic = (2.0 * c['FETCH_SIZE']['mean'] + c['WRITE_SIZE']['mean']) * 1024.0
    if rank == 0:
        out = {
            'metric': 'building-timesteps/sec at 17 bldgs x 65536 envs; HBM GB/s vs roofline',
            'value': world * units_per_step * args.steps / wall_max,
            'unit': 'building-timesteps/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': wall_max / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'citylearn_challenge_2022_phase_all tables (17 buildings, first 720 h) x {E} envs per GPU, '
                                   'cl_step_f32 mode A (one env step per launch, state in HBM, fresh uniform random actions '
                                   'from an 8-tensor ring), env batch sharded over GPUs, no collective',
                       'envs_per_gpu': E, 'buildings': eng.n_bldg, 'launch': 'hipGraph replay' if use_graph else 'eager',
                       'reward': 'RewardFunction'},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
                         'kernel': 'cl_step_envmajor_kernel<20>' if E >= 131072 else 'cl_step_lean_kernel<4, false>',
                         'launch_us': launch_s * 1e6,
                         'algorithmic_bytes_per_unit': bytes_per_unit, 'units_per_launch': units_per_step},
        }
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(spec, tables)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
