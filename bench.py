"""bench.py -- headline benchmark of the MI355X CityLearn step engine.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json `metric`): the 17-building `citylearn_challenge_2022_phase_all` district tables x
65 536 environments per GPU, fp32, one environment step per kernel launch (`cl_step_f32`, mode A: state lives in
HBM, fresh actions every step).  A "step" advances every (env, building) unit by one time step.  The env batch
is sharded across GPUs with no collective on the data path (weak scaling: per-GPU work is fixed).

Timing protocol: W untimed warmup steps, then EXACTLY K steps between barrier + synchronize on both sides, MAX over
ranks -- repeated `--reps` times on the same pre-captured, pre-replayed hipGraphs; the line reports the MEDIAN
repetition (all repetitions are listed in `rep_ms_per_step`).  The kernel duration for the roofline comes from HIP events
recorded on the launch stream around max(K, 2000) consecutive steps of the same loop enqueued behind a lead-in chunk (no host
submission gap inside the bracket); the events around each timed repetition are listed too (`timed_region_event_us_per_step`).

Prints ONE JSON line (rank 0).  `value` = building-timesteps/s over all GPUs with inputs resident in HBM.
`roofline` prices the step kernel against HBM (algorithmic bytes per launch / measured launch duration) at the headline
shape -- whose 58 MB working set sits in the 256 MB Infinity Cache across replays -- and `roofline.hbm_streaming` repeats
the measurement at 17 x 1 048 576 envs (0.66 GB of step traffic per launch, far beyond the cache): the figure that is
bounded by HBM proper.  `cpu_baseline` is the C restatement of the reference arithmetic (oracle/cl_oracle.c, the "port")
timed on this box's host cores on a bounded sample (rank 0, N = 1 only); `cpu_baseline.reference` is the reference's own
`CityLearnEnv.step` as timed by oracle/ref_harness/time_reference.py on the host named there (the reference cannot run
on the GPU box: /root/reference does not travel).
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0           # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
HBM_MEASURED_COPY_GBS = 6290.0  # float4 device copy measured on MI355X, same guide
ENVS_PER_GPU = 65536
STREAMING_ENVS = 1048576        # second roofline entry: working set >> 256 MB Infinity Cache
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
    out = {'value': E * ora.B * n / dt, 'unit': 'building-timesteps/s', 'cores': cores, 'kind': 'port',
           'sample': f'oracle/cl_oracle.c (OpenMP, {cores} threads): 17 buildings x {E} envs x {n} steps in {dt:.1f} s'}
    ref = ROOT / 'profiles' / 'reference_cpu_timing.json'
    if ref.exists():
        # the reference's own CityLearnEnv.step (citylearn.py:978-1056), timed where /root/reference exists
        out['reference'] = json.loads(ref.read_text())
    return out


class Runner:
    """The step loop of one engine as pre-captured hipGraphs: chunk (i0, n) = steps i0 .. i0+n-1 of the action ring / episode."""

    def __init__(self, eng, acts, stream, use_graph: bool):
        self.eng, self.acts, self.stream, self.use_graph = eng, acts, stream, use_graph
        self.T = eng.n_steps - 1                 # an episode of T+1 rows has T transitions
        self.graphs = {}

    def run(self, i0: int, n: int):
        for i in range(i0, i0 + n):
            self.eng.step(self.acts[i % len(self.acts)], i % self.T)

    def chunks(self, i0: int, n: int):
        i = i0
        while i < i0 + n:
            c = min(GRAPH_CHUNK, i0 + n - i)
            yield i, c
            i += c

    def prepare(self, i0: int, n: int):
        """Capture every graph steps [i0, i0+n) need and replay each once, untimed: the first launch of a graph pays its
        upload, which must not land in the timed region."""
        if not self.use_graph:
            return
        for i, c in self.chunks(i0, n):
            key = (i % (len(self.acts) * self.T), c)
            if key not in self.graphs:
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr, stream=self.stream):
                    self.run(i, c)
                gr.replay()
                self.graphs[key] = gr
        self.stream.synchronize()

    def advance(self, i0: int, n: int):
        for i, c in self.chunks(i0, n):
            if self.use_graph:
                self.graphs[(i % (len(self.acts) * self.T), c)].replay()
            else:
                self.run(i, c)


def timed_reps(runner: Runner, warmup: int, steps: int, reps: int, dist, device):
    """`reps` x (exactly `steps` steps between barrier + synchronize): per-repetition (wall seconds, HIP-event seconds)."""
    stream = runner.stream
    out = []
    with torch.cuda.stream(stream):
        runner.run(0, min(max(warmup, 1), 50))       # first touch / module load outside any capture
        stream.synchronize()
        runner.prepare(0, warmup)
        runner.prepare(warmup, steps)
        runner.eng.reset()
        runner.advance(0, warmup)
        stream.synchronize()
        for _ in range(reps):
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            ev0.record(stream)
            runner.advance(warmup, steps)
            ev1.record(stream)
            stream.synchronize()
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            wall = time.perf_counter() - t0
            out.append((wall, ev0.elapsed_time(ev1) / 1e3))
        # Kernel duration for the roofline: HIP events on the launch stream around >= 2000 consecutive steps of the same loop
        # (graphs of GRAPH_CHUNK steps, captured and replayed once beforehand) enqueued behind a lead-in chunk, so that the bracket
        # [ev0, ev1] holds kernel time only.  (Events around a timed repetition also hold the host's graph-submission gap between
        # `ev0` and the first kernel -- ~20 us, i.e. 1 us per step at K = 20 -- and a K-step graph replayed back to back still pays
        # ~5 us per graph boundary: both are launch behaviour of short graphs, not kernel duration.)
        n_k = max(steps, 2000)
        runner.prepare(warmup, n_k)
        runner.advance(warmup, min(GRAPH_CHUNK, n_k))
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record(stream)
        runner.advance(warmup, n_k)
        ev1.record(stream)
        stream.synchronize()
        kernel_s = ev0.elapsed_time(ev1) / 1e3 / n_k
    return out, kernel_s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5000)
    ap.add_argument('--warmup', type=int, default=300)
    ap.add_argument('--reps', type=int, default=5, help='timed repetitions of the K steps; the median is reported')
    ap.add_argument('--envs-per-gpu', type=int, default=ENVS_PER_GPU)
    ap.add_argument('--no-graph', action='store_true', help='launch every step from Python instead of hipGraph replay')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-streaming', action='store_true', help='skip the 17 x 1 048 576 HBM-streaming roofline entry')
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

    from citylearn_amd import load_district
    from citylearn_amd.data import sample_schema
    from citylearn_amd.engine import StepEngine
    from citylearn_amd.parallel import reduce_max_seconds

    spec = load_district(sample_schema('citylearn_challenge_2022_phase_all_720h'))   # 17 buildings, first 720 hours
    tables = spec.episode_tables(0)
    tuning = {k[len('CL_TUNE_'):].lower(): int(v) for k, v in os.environ.items() if k.startswith('CL_TUNE_')}   # e.g. CL_TUNE_ENVMAJOR=2
    use_graph = not args.no_graph

    def measure(E: int, warmup: int, steps: int, reps: int):
        eng = StepEngine(tables, E, device=device, tuning=tuning)
        assert eng.lean
        gen = torch.Generator(device=device).manual_seed(1234 + rank)
        acts = [torch.rand((eng.n_act_cols, E), device=device, generator=gen) * 2 - 1 for _ in range(8)]
        runner = Runner(eng, acts, torch.cuda.Stream(device=device), use_graph)
        rep, kernel_s = timed_reps(runner, warmup, steps, reps, dist, device)
        # MAX over ranks per repetition, then the median repetition
        walls = [reduce_max_seconds(w, dist, device) for w, _ in rep]
        evs = [reduce_max_seconds(e, dist, device) for _, e in rep]
        return eng, walls, evs, reduce_max_seconds(kernel_s, dist, device)

    E = args.envs_per_gpu
    eng, walls, evs, launch_s = measure(E, args.warmup, args.steps, args.reps)
    wall_med = statistics.median(walls)
    units_per_step = eng.n_bldg * E
    bytes_per_unit = eng.algorithmic_bytes_per_unit()
    achieved = units_per_step * bytes_per_unit / launch_s / 1e9
    # <.., true> = plane stores / loads with the non-temporal hint (launches of up to 3 Mi units or from 16 Mi units, csrc/cl_kernels.hip)
    kernel_name = (lambda e: f'cl_step_envmajor_kernel<20, {"true" if (e * 17 <= 3 << 20 or e * 17 >= 16 << 20) else "false"}>' if e >= 131072 else 'cl_step_lean_kernel<4, false, true>')

    traffic, traffic_source = None, None
    pmcs = sorted((ROOT / 'profiles').glob('r*_bench_pmc_summary.json'))        # the newest round's counters of this same workload
    pmc = pmcs[-1] if pmcs else ROOT / 'profiles' / 'none'
    if pmc.exists() and E == ENVS_PER_GPU:
        # HBM bytes per launch from the rocprofv3 --pmc passes of this same workload (separate FETCH_SIZE / WRITE_SIZE
        # runs, KiB units; FETCH_SIZE doubled per the gfx950 wide-load correction of MI355X_MICROARCH.md)
        c = json.loads(pmc.read_text())
        traffic = (2.0 * c['FETCH_SIZE']['mean'] + c['WRITE_SIZE']['mean']) * 1024.0
        traffic_source = pmc.name

    streaming = None
    if not args.no_streaming and E == ENVS_PER_GPU:
        del eng
        torch.cuda.empty_cache()
        s_steps = 20
        eng_s, _, _, launch = measure(STREAMING_ENVS, 5, s_steps, 3)
        s_traffic, s_traffic_source = None, None
        spmc = sorted((ROOT / 'profiles').glob('r*_streaming_pmc_summary.json'))
        if spmc:
            c = json.loads(spmc[-1].read_text())
            s_traffic, s_traffic_source = (2.0 * c['FETCH_SIZE']['mean'] + c['WRITE_SIZE']['mean']) * 1024.0, spmc[-1].name
        a = eng_s.n_bldg * STREAMING_ENVS * eng_s.algorithmic_bytes_per_unit() / launch / 1e9
        streaming = {'workload': f'same tables x {STREAMING_ENVS} envs per GPU ({eng_s.n_bldg * STREAMING_ENVS * eng_s.algorithmic_bytes_per_unit() / 1e6:.0f} MB '
                                 f'of algorithmic traffic per launch, beyond the 256 MB Infinity Cache)',
                     'bound': 'hbm', 'achieved': a, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': a / HBM_PEAK_GBS,
                     'frac_vs_measured_copy': a / HBM_MEASURED_COPY_GBS, 'kernel': kernel_name(STREAMING_ENVS), 'launch_us': launch * 1e6,
                     'units_per_launch': eng_s.n_bldg * STREAMING_ENVS, 'steps': s_steps, 'traffic': s_traffic, 'traffic_source': s_traffic_source,
                     'value': world * eng_s.n_bldg * STREAMING_ENVS / launch}
        n_bldg = eng_s.n_bldg
        del eng_s
        torch.cuda.empty_cache()
    else:
        n_bldg = eng.n_bldg

    if rank == 0:
        out = {
            'metric': 'building-timesteps/sec at 17 bldgs x 65536 envs; HBM GB/s vs roofline',
            'value': world * units_per_step * args.steps / wall_med,
            'unit': 'building-timesteps/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': wall_med / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'citylearn_challenge_2022_phase_all tables (17 buildings, first 720 h) x {E} envs per GPU, '
                                   'cl_step_f32 mode A (one env step per launch, state in HBM, fresh uniform random actions '
                                   'from an 8-tensor ring), env batch sharded over GPUs, no collective',
                       'envs_per_gpu': E, 'buildings': n_bldg, 'launch': 'hipGraph replay' if use_graph else 'eager',
                       'reward': 'RewardFunction', 'reps': args.reps, 'statistic': 'median of reps (each: MAX over ranks)'},
            'rep_ms_per_step': [w / args.steps * 1e3 for w in walls],
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic, 'traffic_source': traffic_source,
                         'frac_vs_measured_copy': achieved / HBM_MEASURED_COPY_GBS,
                         'kernel': kernel_name(E), 'launch_us': launch_s * 1e6,
                         'launch_us_how': 'HIP events on the launch stream around max(K, 2000) consecutive steps (pre-replayed 100-step hipGraphs) '
                                          'enqueued behind a lead-in chunk: kernel time only',
                         'timed_region_event_us_per_step': [e / args.steps * 1e6 for e in evs],
                         'algorithmic_bytes_per_unit': bytes_per_unit, 'units_per_launch': units_per_step,
                         'note': 'working set (state 13 MB + outputs 9 MB + action ring 36 MB) fits the 256 MB Infinity Cache: see hbm_streaming '
                                 'for the HBM-resident figure',
                         'hbm_streaming': streaming},
        }
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(spec, tables)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
