"""bench.py -- benchmark of the MI355X CityLearn step engine (BASELINE.json's metric and configs).

    python bench.py --gpus N --steps K --warmup W [--config headline|C2|C3|C4|C4-lean|C5]

`--gpus N` with N > 1 needs no launcher: when RANK / WORLD_SIZE are not in the environment this process starts the N ranks
itself (one process per GPU, `citylearn_amd.parallel.launch_ranks`), forwards rank 0's JSON line and exits with the ranks'
exit code.  Under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...` (RANK etc. set) it
is one of the ranks.  Ranks use RCCL (`nccl`) for the barrier and the MAX-over-ranks of the timing only: the env batch is
sharded across GPUs and no collective sits on the data path (weak scaling: per-GPU work is fixed).

Configs (`--config`, default `headline`):
  headline  BASELINE.json `metric`: 2022_phase_all tables (17 buildings, the full year: T = 8 760 rows) x 65 536 envs per GPU, one env step per launch (mode A)
  C2        the same tables x 4 096 envs per GPU
  C3        2023 phase-2 schema (3 buildings: outage path, partial-load cooling, DHW tank, battery) x 65 536 envs per GPU; a step =
            the energy step AND the LSTM indoor-temperature stage with its ComfortReward epilogue (what CityLearnEnv.step runs there)
  C4        synthetic 1024-building district (2020 climate-zone-1 device set: heat pump, heater, two tanks, battery; parameters
            jittered) x 1024 envs per GPU -- the per-GPU shard of the 8192-env config; C4-lean: battery + PV device set
  C4-B / C4-lean-B  the same districts in mode B: 24 fused env steps per launch through the building-chunked cl_rollout_f32 (round 5)
  C5        fused 24-step day rollout per launch with the on-device Philox random policy, 17 buildings x 32 768 envs per GPU (the
            per-GPU shard of 262 144 envs on 8 GPUs); a "step" of the line is one 24-step launch

Timing protocol: W untimed warmup steps, then EXACTLY K steps bracketed by barrier + synchronize on both sides (every rank reads its
clock between its own two synchronizes, the closing barrier follows), MAX over ranks --
repeated `--reps` times on the same pre-captured, pre-replayed hipGraphs; the line reports the MEDIAN repetition (all repetitions
in `rep_ms_per_step`, every rank's median in `rank_ms_per_step`).  The kernel duration behind `roofline` comes from HIP events
recorded on the launch stream around max(K, 2000) consecutive steps of the same loop enqueued behind a lead-in chunk (no host
submission gap inside the bracket); `roofline.kernel` is what the library reports having launched (`cl_tuning.kernel_name`).

Prints ONE JSON line (rank 0).  `value` = building-timesteps/s over all GPUs with inputs resident in HBM.  `roofline.traffic` of the headline
line (N = 1) and of its `hbm_streaming` entry is measured by the run itself: two child runs each of this script under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (`_live_traffic`;
`--no-traffic-pass` skips them and falls back to the newest matching summary under profiles/); the other configs read that summary.  `cpu_baseline`
(headline, N = 1) is the C restatement of the reference arithmetic (oracle/cl_oracle.c, the "port") timed on this box's host
cores -- all cores, and one core as `cpu_baseline.one_core`; `cpu_baseline.reference` is the reference's own `CityLearnEnv.step`
timed in the same run on the same host by oracle/ref_harness/time_reference.py (usable-cores processes x 1000 steps of 2022_phase_all: SURVEY 8d-ii)
from the staging `oracle/_ref/reference` that build() makes (git-ignored; travels with the snapshot like the built .so files).

Test hooks (environment): CL_BENCH_OVERSUBSCRIBE=1 maps rank r to device r mod (visible devices) so that `--gpus 2` can be
exercised on a 1-GPU box (the ranks then share a GPU: control plane over gloo because RCCL refuses two ranks per device, and the
line says `"oversubscribed": true` -- not a scaling measurement); CL_BENCH_DRY_RUN=1 skips all GPU work (launcher, rendezvous and
aggregation on CPU: tests/test_distributed.py; refused when a GPU is visible); CL_BENCH_FORCE_DIST=1 brings up the process group even for one rank; CL_BENCH_CONTROL=nccl|gloo
picks the control plane's backend (default: RCCL with one rank per GPU; if RCCL cannot come up the barrier falls back to gloo and the line says
so in `control_fallback` -- CL_BENCH_STRICT_RCCL=1 makes that fatal instead); CL_BENCH_EXTRA_CONFIGS=1 runs the N > 1 line's `extra_configs` (C4, C4-lean, C5 in the same run)
whatever the rank count; CL_BENCH_NO_PIN=1 leaves the ranks' CPU affinity alone.
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0           # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
HBM_MEASURED_COPY_GBS = 6290.0  # float4 device copy measured on MI355X, same guide
VALU_LANES = 256 * 4 * 16       # 256 CUs x 4 SIMDs x 16 lanes
VALU_CLOCK_GHZ = 2.4            # MI355X peak engine clock
ENVS_PER_GPU = 65536
STREAMING_ENVS = 1048576        # second roofline entry of the headline: working set >> 256 MB Infinity Cache
GRAPH_CHUNK = 100
ROUND_PREFIX = 'r06'             # profiles/ files this build's lines may cite (scripts/profile_round.sh writes them)
METRIC = 'building-timesteps/sec at 17 bldgs x 65536 envs; HBM GB/s vs roofline'
PRECISIONS = {'chain': 'chain', 'fp32': False, 'f64': True}      # --precision -> StepEngine(f64_maps=...)
CONFIGS = ('headline', 'C2', 'C3', 'C3-6', 'C4', 'C4-lean', 'C5', 'T9', 'C4-B', 'C4-lean-B')


# --------------------------------------------------------------------------------------------------- CPU baseline
def usable_cores() -> int:
    """Host cores this process may really use: the affinity mask, cut by a cgroup CPU quota if there is one (a box that shows 256
    logical CPUs but grants a few through cpu.max throttles 256 spinning OpenMP threads to a crawl: 116 ms per oracle step)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = Path('/sys/fs/cgroup/cpu.max').read_text().split()[:2]                     # cgroup v2
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            quota = int(Path('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read_text())                  # cgroup v1
            period = int(Path('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read_text())
            if quota > 0:
                n = min(n, max(1, quota // period))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(spec, tables, seconds: float = 10.0) -> dict:
    """Time oracle/cl_oracle.c (double-precision port of the reference arithmetic, OpenMP over envs) on a bounded sample of the
    headline workload: the host's usable cores (thread count picked by a short probe among usable, 64, 16: the one that is
    fastest on this box), then ONE core (the "fair CPU" line of SURVEY 8d)."""
    import ctypes
    import numpy as np
    from oracle.c_oracle import COracle
    cores = usable_cores()
    os.environ.setdefault('OMP_NUM_THREADS', str(cores))
    omp = ctypes.CDLL('libgomp.so.1')

    def run(threads: int, E: int, budget: float):
        omp.omp_set_num_threads(threads)
        ora = COracle(spec, tables, E)
        rng = np.random.RandomState(0)
        acts = [rng.uniform(-1, 1, size=(ora.n_act_cols, E)).astype(np.float32) for _ in range(4)]
        for t in range(8):                                  # thread pool start-up, first touch
            ora.step(acts[t % 4], t)
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < budget:
            for _ in range(5):
                ora.step(acts[n % 4], 1 + n % (ora.T - 2))
                n += 1
        dt = time.perf_counter() - t0
        return E * ora.B * n / dt, f'oracle/cl_oracle.c (OpenMP, {threads} thread{"s" if threads > 1 else ""}): 17 buildings x {E} envs x {n} steps in {dt:.1f} s'

    E = 16384
    probe = {th: run(th, E, 1.0)[0] for th in sorted({cores, *(c for c in (64, 16) if c < cores)})}
    threads = max(probe, key=probe.get)
    v, sample = run(threads, E, seconds)
    v1, sample1 = run(1, 256, seconds / 2)
    host = {'logical_cpus': os.cpu_count(), 'usable_cores': cores, 'thread_probe': {str(k): p for k, p in probe.items()}}
    port = {'value': v, 'unit': 'building-timesteps/s', 'cores': threads, 'kind': 'port', 'sample': sample, 'host': host,
            'one_core': {'value': v1, 'unit': 'building-timesteps/s', 'cores': 1, 'kind': 'port', 'sample': sample1}}
    # The stated baseline is the REFERENCE's own step (north_star: "next to the reference CPU step timed on the GPU box's own host cores");
    # the C restatement of its arithmetic rides along as `port` (what a compiled CPU implementation of the same path reaches).
    out = reference_cpu_baseline(cores)
    out['port'] = port
    if out.get('value') is None:
        # no reference timing at all (staging absent AND no committed file): the port is the only CPU number this run has -- say so
        out.update({'value': v, 'unit': 'building-timesteps/s', 'cores': threads, 'kind': 'port', 'sample': sample,
                    'note': 'reference staging absent: top level falls back to the C port'})
    return out


def reference_cpu_baseline(cores: int, steps: int = 1000, timeout: float = 600.0) -> dict:
    """The REFERENCE's own `CityLearnEnv.step` (citylearn.py:978-1056) timed on THIS host in THIS run: `cores` independent processes
    (the path has no intra-step threading), each stepping its own citylearn_challenge_2022_phase_all env for `steps` steps --
    oracle/ref_harness/time_reference.py in a subprocess, importing the staging `oracle/_ref/reference` that `__graft_entry__.build()`
    made from /root/reference (git-ignored; travels to the GPU box like the built .so files).  When the staging is absent or fails,
    the build container's committed timing is attached instead and says so."""
    import subprocess
    staged = ROOT / 'oracle' / '_ref' / 'reference'
    committed = ROOT / 'profiles' / 'reference_cpu_timing.json'
    err = None
    sys.path.insert(0, str(ROOT / 'oracle' / 'ref_harness'))
    import stage_reference                                    # (measurement side: only this leg touches oracle/)
    if (staged / 'MANIFEST.json').is_file() and not stage_reference.verify(staged):
        err = 'oracle/_ref/reference does not match its manifest (re-run __graft_entry__.build() where /root/reference exists)'
    elif (staged / 'MANIFEST.json').is_file():
        cmd = [sys.executable, str(ROOT / 'oracle' / 'ref_harness' / 'time_reference.py'), '--root', str(staged), '--c1-budget', '60',
               '--procs', str(cores), '--steps', str(steps), '--out', '-']
        try:
            t0 = time.perf_counter()
            p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=str(ROOT),
                               env={**os.environ, 'OMP_NUM_THREADS': '1', 'MKL_NUM_THREADS': '1', 'HIP_VISIBLE_DEVICES': ''})
            lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{')]
            if p.returncode == 0 and lines:
                ref = json.loads(lines[-1])
                ref['measured'] = 'live, in this bench run, on this host'
                ref['seconds_including_env_construction'] = round(time.perf_counter() - t0, 1)
                try:
                    import torch
                    ref['host_gpu'] = torch.cuda.get_device_name(0) if torch.cuda.is_available() else None
                    ref['host_gpu_arch'] = getattr(torch.cuda.get_device_properties(0), 'gcnArchName', None) if torch.cuda.is_available() else None
                except Exception:
                    ref['host_gpu'] = None
                return ref
            err = f'time_reference.py rc {p.returncode}: {(p.stderr or p.stdout)[-400:]}'
        except subprocess.TimeoutExpired:
            err = f'time_reference.py exceeded {timeout:.0f} s'
    else:
        err = 'oracle/_ref/reference not staged (run __graft_entry__.build() where /root/reference exists)'
    ref = json.loads(committed.read_text()) if committed.exists() else {'kind': 'reference', 'value': None}
    ref['measured'] = 'NOT in this run: committed timing from the build container (profiles/reference_cpu_timing.json)'
    ref['live_error'] = err
    return ref


def dropin_timing(device: str, cpu: dict) -> dict:
    """BASELINE config 1 through the DROP-IN class: `citylearn_amd.CityLearnEnv` (lists in, lists out, one district, the reference's reset / step
    surface; citylearn.py:52, 978-1056) over 2022_phase_1's full 8 759-step episode with the survey's action stream (RandomState(0), one uniform
    draw per building and step), wall clock around the step loop -- beside the reference's own time for the same episode on this host
    (`cpu_baseline.c1_single_process`).  A latency number (one 4-env launch + three small device-to-host copies per step), not a throughput one."""
    import numpy as np
    try:
        from citylearn_amd.citylearn import CityLearnEnv
        schema = ROOT / 'tests' / 'golden' / 'g2022_p1_year' / 'dataset' / 'schema.json'          # (a data fixture: the 2022_phase_1 CSVs, 5 x 8 760 rows)
        t0 = time.perf_counter()
        env = CityLearnEnv(str(schema), device=device)
        env.reset()
        built = time.perf_counter() - t0
        rng = np.random.RandomState(0)
        n_b = len(env.action_names)
        acts = [[list(rng.uniform(-1, 1, size=len(names))) for names in env.action_names] for _ in range(env.time_steps - 1)]   # (generated outside the clock, like time_reference.py)
        n, t0 = 0, time.perf_counter()
        for a in acts:
            env.step(a)
            n += 1
        dt = time.perf_counter() - t0
        assert env.terminated
        out = {'what': 'citylearn_amd.CityLearnEnv (single district, lists in / lists out, default arguments: CLD_F64_CHAIN + CLD_CHECK) on citylearn_challenge_2022_phase_1, '
                       f'{n_b} buildings, full episode', 'steps': n, 'seconds': dt, 'us_per_step': dt / n * 1e6, 'value': n_b * n / dt, 'unit': 'building-timesteps/s',
               'construction_and_reset_seconds': built}
        c1 = (cpu or {}).get('c1_single_process') or {}
        if c1.get('seconds') and c1.get('steps'):
            out['reference_same_episode'] = {'seconds': c1['seconds'], 'steps': c1['steps'], 'host': 'this host, this run' if 'live' in str((cpu or {}).get('measured', '')) else 'committed timing'}
            out['speedup_vs_reference'] = (c1['seconds'] / c1['steps']) / (dt / n)
        return out
    except Exception as e:                                                                       # a missing fixture must not cost the run its line
        return {'error': f'{type(e).__name__}: {e}'}


# --------------------------------------------------------------------------------------------------- step loop as hipGraphs
class Runner:
    """The step loop of one workload as pre-captured hipGraphs: chunk (i0, n) = steps i0 .. i0+n-1; `step_fn(i)` enqueues step i
    on the current stream and depends on i only through i mod `period`."""

    def __init__(self, step_fn, period: int, stream, use_graph: bool, flush_fn=None):
        self.step_fn, self.period, self.stream, self.use_graph = step_fn, period, stream, use_graph
        self.flush_fn = flush_fn           # end of a step sequence (a captured chunk): e.g. the deferred finish of the C4 shard's district sums
        self.graphs = {}

    def run(self, i0: int, n: int):
        for i in range(i0, i0 + n):
            self.step_fn(i)
        if self.flush_fn is not None:
            self.flush_fn()

    def chunks(self, i0: int, n: int):
        i = i0
        while i < i0 + n:
            c = min(GRAPH_CHUNK, i0 + n - i)
            yield i, c
            i += c

    def prepare(self, i0: int, n: int):
        """Capture every graph steps [i0, i0+n) need and replay each once, untimed: the first launch of a graph pays its
        upload, which must not land in the timed region."""
        import torch
        if not self.use_graph:
            return
        for i, c in self.chunks(i0, n):
            key = (i % self.period, c)
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
                self.graphs[(i % self.period, c)].replay()
            else:
                self.run(i, c)

    def plan(self, i0: int, n: int):
        """`advance(i0, n)` with the chunk arithmetic and the graph look-ups done beforehand: a callable for the timed region."""
        if not self.use_graph:
            return lambda: self.advance(i0, n)
        replays = [self.graphs[(i % self.period, c)].replay for i, c in self.chunks(i0, n)]
        if len(replays) == 1:
            return replays[0]
        return lambda: [r() for r in replays]


def _barrier(dist):
    """The bracket's barrier, enqueued on the DEFAULT stream -- never on the stream the graphs are captured on.  torch runs a blocking
    collective on the caller's current stream and its watchdog thread keeps polling the collective's end event for a while after it has
    completed; HIP refuses `hipEventQuery` on an event whose last-recorded stream is capturing (hipErrorCapturedEvent), the watchdog
    throws and aborts the process.  Seen as 1 abort in ~10 lone-rank RCCL runs when the barrier shared the launch stream with the
    captures that follow it (`gpurun` runs 36 / 37 of round 3); with gloo there is no such event, which is why no earlier test saw it."""
    import torch
    with torch.cuda.stream(torch.cuda.default_stream()):
        dist.barrier()
    torch.cuda.default_stream().synchronize()


def timed_reps(runner: Runner, reset_fn, warmup: int, steps: int, reps: int, dist, kernel_steps: int):
    """`reps` x (exactly `steps` steps between barrier + synchronize): per-repetition (wall seconds, HIP-event seconds), and the
    kernel time per step from HIP events around `kernel_steps` consecutive steps behind a lead-in chunk."""
    import torch
    stream = runner.stream
    out = []
    with torch.cuda.stream(stream):
        runner.run(0, min(max(warmup, 1), 50))       # first touch / module load outside any capture
        stream.synchronize()
        runner.prepare(0, warmup)
        runner.prepare(warmup, steps)
        reset_fn()
        runner.advance(0, warmup)
        stream.synchronize()
        go = runner.plan(warmup, steps)
        for _ in range(reps):
            if dist is not None:
                _barrier(dist)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            go()
            torch.cuda.synchronize()                 # (the device-wide wait covers the launch stream: a stream.synchronize() in front of it was a second
                                                     #  host round trip inside the region -- 8.96 - 9.29 -> 8.89 - 8.95 us per step at the driver's K = 20)
            wall = time.perf_counter() - t0          # this rank's K steps, synchronize to synchronize; the line reports the MAX over ranks
            if dist is not None:
                _barrier(dist)                       # the closing barrier of the bracket: after the clock is read -- a 30 us RCCL barrier inside
                                                     # a 20-step region would bill every rank 1.5 us per step for the collective's own latency
            # the same K steps once more between HIP events (diagnostic `timed_region_event_us_per_step`), OUTSIDE the wall-clock region: two
            # event records inside a 20-step region cost it ~3 us = 2 % (scripts/r05_k20_overhead.py, profiles/r05q_k20_overhead.log)
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record(stream)
            runner.advance(warmup, steps)
            ev1.record(stream)
            stream.synchronize()
            out.append((wall, ev0.elapsed_time(ev1) / 1e3))
        # Kernel duration for the roofline: HIP events on the launch stream around `kernel_steps` consecutive steps of the same loop
        # (graphs of GRAPH_CHUNK steps, captured and replayed once beforehand) enqueued behind a lead-in chunk, so that the bracket
        # [ev0, ev1] holds kernel time only.  (Events around a timed repetition also hold the host's graph-submission gap between
        # `ev0` and the first kernel -- ~20 us, i.e. 1 us per step at K = 20 -- and a K-step graph replayed back to back still pays
        # ~5 us per graph boundary: both are launch behaviour of short graphs, not kernel duration.)
        n_k = kernel_steps
        runner.prepare(warmup, n_k)
        runner.advance(warmup, min(GRAPH_CHUNK, n_k))
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record(stream)
        runner.advance(warmup, n_k)
        ev1.record(stream)
        stream.synchronize()
        kernel_s = ev0.elapsed_time(ev1) / 1e3 / n_k
    return out, kernel_s


# --------------------------------------------------------------------------------------------------- workloads
def _pmc_traffic(pattern: str, kernels: str):
    """HBM bytes per launch from the newest rocprofv3 --pmc summary under profiles/ matching `pattern` (separate FETCH_SIZE /
    WRITE_SIZE passes, KiB units; FETCH_SIZE doubled per the gfx950 wide-load correction of MI355X_MICROARCH.md) -- only if
    the summary was collected on the kernel this run launched (`_kernel.kernel` must contain its name)."""
    # (only summaries of the CURRENT round: a stale file of an earlier round matched by glob is the cheapest way for a line to be wrong --
    #  VERDICT r05 item 10; earlier rounds' files live under profiles/archive/)
    files = sorted(f for f in (ROOT / 'profiles').glob(pattern) if f.name.startswith(ROUND_PREFIX))
    for f in reversed(files):
        c = json.loads(f.read_text())
        seen = c.get('_kernel', {}).get('kernel', '')
        if 'FETCH_SIZE' in c and 'WRITE_SIZE' in c and any(k and k in seen for k in kernels.split('+')):
            return (2.0 * c['FETCH_SIZE']['mean'] + c['WRITE_SIZE']['mean']) * 1024.0, f.name
    return None, None


_LIVE_TRAFFIC_DISABLED = None          # why the counter passes were given up in this run (a failing / hanging rocprofv3 is tried ONCE: 60 s at most)


def _live_traffic(kernels: str, extra_args, timeout: float = 60.0, steps: int = 200, warmup: int = 50):
    """HBM bytes per launch of `kernels`' first kernel MEASURED IN THIS RUN: two child runs of this script under
    `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes, counter collection only -- no tracing domain beside it -- as
    /opt/skills/guides/MI355X_MICROARCH.md prescribes; KiB units, FETCH_SIZE doubled: its gfx950 wide-load correction), eager launches
    of the same workload, the dispatches with the kernel's largest grid averaged.  Returns (bytes, source string) or (None, reason)."""
    import csv
    import shutil
    import subprocess
    import tempfile
    global _LIVE_TRAFFIC_DISABLED
    if _LIVE_TRAFFIC_DISABLED:
        return None, _LIVE_TRAFFIC_DISABLED
    exe = shutil.which('rocprofv3') or ('/opt/rocm/bin/rocprofv3' if Path('/opt/rocm/bin/rocprofv3').exists() else None)
    if exe is None:
        return None, 'rocprofv3 not found'
    # never nest: a run that is itself being profiled (rocprofv3 exports ROCPROF* / ROCP_* / ROCPROFILER_* and preloads its tool library) would
    # hand its tracing tool to the children -- counters beside a tracing domain is the one combination the profiling guide forbids
    profiled = [k for k in os.environ if k.startswith(('ROCPROF', 'ROCP_', 'ROCPROFILER'))] or 'rocprofiler' in os.environ.get('LD_PRELOAD', '')
    if profiled:
        return None, 'this run is itself under a rocprofiler tool: no nested counter pass'
    needle = (kernels or '').split('+')[0]
    means, n_disp = {}, 0
    for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
        out = tempfile.mkdtemp(prefix=f'cl_pmc_{counter}_', dir='/tmp')
        cmd = [exe, '--pmc', counter, '--output-format', 'csv', '-d', out, '-o', 'run', '--', sys.executable, str(Path(__file__).resolve()),
               '--steps', str(steps), '--warmup', str(warmup), '--reps', '1', '--no-cpu-baseline', '--no-graph', '--no-streaming', '--no-traffic-pass', '--no-side-entries', *extra_args]
        try:
            p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd='/tmp', env={**os.environ, 'TMPDIR': '/tmp'})
            files = list(Path(out).rglob('*counter_collection.csv'))
            if p.returncode != 0 or not files:
                _LIVE_TRAFFIC_DISABLED = f'rocprofv3 --pmc {counter}: rc {p.returncode} {(p.stderr or "")[-200:]}'
                return None, _LIVE_TRAFFIC_DISABLED
            rows = []
            for f in files:
                with open(f, newline='') as fh:
                    rows += [r for r in csv.DictReader(fh) if needle and needle in r['Kernel_Name'] and r['Counter_Name'] == counter]
            if not rows:
                return None, f'no dispatch of {needle} in the {counter} pass'
            biggest = max(int(r['Grid_Size']) for r in rows)
            vals = [float(r['Counter_Value']) for r in rows if int(r['Grid_Size']) == biggest]
            means[counter], n_disp = sum(vals) / len(vals), len(vals)
        except subprocess.TimeoutExpired:
            _LIVE_TRAFFIC_DISABLED = f'rocprofv3 --pmc {counter} exceeded {timeout:.0f} s'
            return None, _LIVE_TRAFFIC_DISABLED
        finally:
            shutil.rmtree(out, ignore_errors=True)
    return (2.0 * means['FETCH_SIZE'] + means['WRITE_SIZE']) * 1024.0, \
        f'measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes, {n_disp} dispatches of {needle} (2 x FETCH + WRITE, KiB)'


class StepWorkload:
    """Mode A: one env step per launch on a district's tables (`cl_step_f32`), fresh actions from an 8-tensor ring."""

    dtype = 'f32'

    def __init__(self, name: str, spec, E: int, device: str, rank: int, tuning: dict, what: str, lstm: bool = False, f64: bool = False, kpi: bool = False):
        import torch
        from citylearn_amd.engine import StepEngine
        self.name, self.what, self.E, self.device = name, what, E, device
        self.spec = spec
        self.tables = spec.episode_tables(0)
        # (CL_BENCH_NO_PITCH=1: rows exactly n_env floats apart even where the engine would pad them -- the A/B of cl_dims.env_pitch)
        self.eng = StepEngine(self.tables, E, device=device, tuning=tuning, detail='min' if lstm else False, f64_maps=f64, kpi=kpi,
                              env_pitch=E if os.environ.get('CL_BENCH_NO_PITCH') == '1' else None)
        if kpi:
            self.what += '; CLD_KPI: streaming KPI accumulators of evaluate() updated every step (mode A-kpi)'
        if f64 == 'chain':
            self.what += '; CLD_F64_CHAIN: battery soc chain in float64, degraded capacity carried as the capacity loss'
        elif f64:
            self.what += '; CLD_F64_MAPS: battery map in the reference\'s mixed float64 / float32 precision'
        self.eng.trace_kernels()
        self.stage = None
        if lstm:
            from citylearn_amd.dynamics import LSTMStage
            a = dict(spec.reward_function.get('attributes') or {})
            self.stage = LSTMStage(spec, self.tables, self.eng, a.get('band'), a.get('lower_exponent') or 2.0, a.get('higher_exponent') or 2.0)
        low, high = spec.action_limits()
        lo, hi = torch.from_numpy(low).to(device), torch.from_numpy(high).to(device)
        gen = torch.Generator(device=device).manual_seed(1234 + rank)
        self.acts = [lo[:, None] + torch.rand((self.eng.n_act_cols, E), device=device, generator=gen) * (hi - lo)[:, None] for _ in range(8)]
        self.T = self.eng.n_steps - 1                 # an episode of T+1 rows has T transitions
        self.period = len(self.acts) * self.T
        self.units_per_step = self.eng.n_bldg * E
        self.kernels = None
        self.lstm_kernels = None

    def step_fn(self, i: int):
        t = i % self.T
        self.eng.step(self.acts[i % len(self.acts)], t)
        if self.kernels is None:
            self.kernels = self.eng.last_kernels
        if self.stage is not None:
            self.stage.step(t)
            if t >= 13 and self.lstm_kernels is None:
                self.lstm_kernels = self.eng.last_kernels

    def reset(self):
        self.eng.reset()
        if self.stage is not None:
            self.stage.reset()

    def flush(self):
        """End of a step sequence (every captured chunk of <= GRAPH_CHUNK steps ends with it, inside the timed region): brings out_env up
        to date where the district sums are finished deferred (C4 shard, `cl_tuning.finish = 3`); nothing to enqueue otherwise."""
        self.eng.finish()

    def bytes_per_unit(self) -> float:
        return self.eng.algorithmic_bytes_per_unit()

    def roofline(self, launch_s: float) -> dict:
        bpu = self.bytes_per_unit()
        achieved = self.units_per_step * bpu / launch_s / 1e9
        r = {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
             'frac_vs_measured_copy': achieved / HBM_MEASURED_COPY_GBS, 'kernel': self.kernels, 'launch_us': launch_s * 1e6,
             'algorithmic_bytes_per_unit': bpu, 'units_per_launch': self.units_per_step}
        if self.stage is not None:
            # C3: the step is two launches and the LSTM stage dominates it; it is bound by vector-ALU issue, its transcendentals first
            # (12 window steps x 2 layers x 16 units x 10 exp / rcp per unit-step; quarter-rate instructions: 4 lanes per SIMD and cycle)
            trans = 12 * 2 * 16 * (7 if '<32,' in (self.lstm_kernels or '') else 10)
            peak = VALU_LANES / 4 * VALU_CLOCK_GHZ             # G transcendentals / s
            ach = self.units_per_step * trans / launch_s / 1e9
            r = {'bound': 'valu', 'achieved': ach, 'peak': peak, 'unit': 'G transcendental op/s', 'frac': ach / peak,
                 'kernel': f'{self.kernels}+{self.lstm_kernels}', 'launch_us': launch_s * 1e6, 'units_per_launch': self.units_per_step,
                 'transcendentals_per_unit': trans, 'traffic': None,
                 'note': 'launch_us = energy step + LSTM stage of one env step; the LSTM kernel re-runs the 12-step lookback window of a 2 x 16-unit '
                         'LSTM per (env, building) and step (building.py:3000-3078), ~77 kFLOP on the matrix cores beside the activations that bound it; '
                         'peak = 256 CUs x 4 SIMDs x 4 quarter-rate lanes x 2.4 GHz',
                 'energy_step_hbm': {'algorithmic_bytes_per_unit': bpu, 'note': 'see profiles/ kernel stats for the split of launch_us'}}
        return r


class RolloutWorkload:
    """Mode B (config C5): K fused steps per launch, state in registers, on-device Philox policy (`cl_rollout_f32`)."""

    dtype = 'f32'

    def __init__(self, name: str, spec, E: int, K: int, device: str, rank: int, world: int, tuning: dict, what: str, f64=False, valu_per_unit_step: float = 100.0,
                 valu_source: str = 'profiles/archive/r02b_rollout_pmc_by_kernel.jsonl: 100 VALU instructions per unit-step at two envs per lane, 101 at one',
                 valu_chain: float = None):
        import torch
        from citylearn_amd.engine import StepEngine
        self.name, self.what, self.E, self.K, self.device = name, what, E, K, device
        self.spec = spec
        self.tables = spec.episode_tables(0)
        self.eng = StepEngine(self.tables, E, device=device, tuning=tuning, env_offset=rank * E, f64_maps=f64)     # disjoint Philox streams per shard
        self.eng.trace_kernels()
        low, high = spec.action_limits()
        self.eng.set_action_limits(low, high)
        self.ret = torch.zeros(E, device=device)
        self.n_windows = (self.eng.n_steps - 1) // K            # whole K-step windows of the episode
        self.period = self.n_windows
        self.units_per_step = self.eng.n_bldg * E * K
        self.kernels = None
        self.inst, self.inst_source, self.inst_chain = valu_per_unit_step, valu_source, valu_chain

    def step_fn(self, i: int):
        w = i % self.n_windows
        self.eng.rollout(self.K, seed=5 + i % self.period, ret_env=self.ret, t0=w * self.K)
        if self.kernels is None:
            self.kernels = self.eng.last_kernels

    def reset(self):
        self.eng.reset()
        self.ret.zero_()

    def bytes_per_unit(self) -> float:
        return self.eng.algorithmic_bytes_per_unit() / self.K

    def roofline(self, launch_s: float) -> dict:
        # VALU-issue bound: lane-instructions per (env, building, step) from the SQ counters of this kernel (`inst_source`)
        inst = self.inst_chain if (self.eng.f64_chain and self.inst_chain) else self.inst
        peak = VALU_LANES * VALU_CLOCK_GHZ                      # G lane-instructions / s
        ach = self.units_per_step * inst / launch_s / 1e9
        if self.eng.f64_chain and not self.inst_chain:
            # the chain's float64 instructions issue at half rate and its count per unit-step was not collected: no VALU fraction claimed
            return {'bound': 'valu', 'achieved': None, 'peak': peak, 'unit': 'G lane-instructions/s', 'frac': None, 'kernel': self.kernels,
                    'launch_us': launch_s * 1e6, 'units_per_launch': self.units_per_step, 'traffic': None,
                    'hbm_bytes_per_unit_step': self.bytes_per_unit(),
                    'note': 'CLD_F64_CHAIN fused rollout: the fp32 kernel of this config is the one priced against the VALU-issue bound (profiles/r05i: 61.8 vs 44.5 us '
                            'per 24-step launch at 17 x 32 768)'}
        return {'bound': 'valu', 'achieved': ach, 'peak': peak, 'unit': 'G lane-instructions/s', 'frac': ach / peak, 'kernel': self.kernels,
                'launch_us': launch_s * 1e6, 'units_per_launch': self.units_per_step, 'valu_instructions_per_unit_step': inst,
                'valu_instructions_source': self.inst_source, 'traffic': None, 'hbm_bytes_per_unit_step': self.bytes_per_unit(),
                'note': f'one launch = {self.K} env steps with state in registers; HBM sees state once per launch, so the bound is vector-ALU issue '
                        '(SURVEY 8d): achieved = units x VALU instructions per unit-step (SQ_INSTS_VALU, profiles/) / launch time, '
                        'peak = 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz'}


def load_c2_spec(hours: int):
    from citylearn_amd import load_district
    from citylearn_amd.data import sample_schema
    return load_district(sample_schema(f'citylearn_challenge_2022_phase_all_{hours}h'))


def build_workload(cfg: str, E: int, device: str, rank: int, world: int, tuning: dict, f64: bool = False, kpi: bool = False, hours: int = 8760):
    from citylearn_amd import load_district
    from citylearn_amd.data import sample_schema
    if cfg in ('headline', 'C2', 'C5'):
        # 17 buildings; the whole year (T = 8 760 table rows) unless --table-hours 720 asks for the 720-hour cut of rounds 1 - 4
        spec = load_district(sample_schema(f'citylearn_challenge_2022_phase_all_{hours}h'))
        span = 'the full year, 8 760 h' if hours == 8760 else f'first {hours} h'
        if cfg == 'C5':
            return RolloutWorkload(cfg, spec, E, 24, device, rank, world, tuning,
                                   f'citylearn_challenge_2022_phase_all tables (17 buildings, {span}) x {E} envs per GPU, cl_rollout_f32 mode B: '
                                   '24 fused env steps per launch, state in registers, on-device Philox4x32-10 uniform random policy; env batch sharded '
                                   'over GPUs (8 x 32 768 = the 262 144 envs of BASELINE config 5), no collective', f64=f64)
        return StepWorkload(cfg, spec, E, device, rank, tuning,
                            f'citylearn_challenge_2022_phase_all tables (17 buildings, {span}) x {E} envs per GPU, '
                            'cl_step_f32 mode A (one env step per launch, state in HBM, fresh uniform random actions '
                            'from an 8-tensor ring), env batch sharded over GPUs, no collective', f64=f64, kpi=kpi)
    if cfg == 'C3':
        spec = load_district(sample_schema('citylearn_challenge_2023_phase_2_local_evaluation_720h'))
        return StepWorkload(cfg, spec, E, device, rank, tuning,
                            f'citylearn_challenge_2023_phase_2_local_evaluation (3 buildings: power outages, partial-load cooling, DHW tank, battery; '
                            f'first 720 h) x {E} envs per GPU; one step = cl_step_f32 (energy step + the delivered-demand planes the stage reads) + cl_lstm_step_f32 (LSTM indoor '
                            'temperature + ComfortReward): the whole CityLearnEnv.step of this schema', lstm=True, f64=f64, kpi=kpi)
    if cfg == 'C3-6':
        # the six-building 2023 district (SURVEY 8d lists it beside the three-building one): the first 96 hours that ship as the parity fixture s_2023_p3
        spec = load_district(str(ROOT / 'tests' / 'golden' / 's_2023_p3' / 'dataset' / 'schema.json'))
        return StepWorkload(cfg, spec, E, device, rank, tuning,
                            f'citylearn_challenge_2023_phase_3_1 (6 buildings: power outages, partial-load cooling, DHW tank, battery; first 96 h) x {E} envs per GPU; '
                            'one step = cl_step_f32 + cl_lstm_step_f32 (LSTM indoor temperature + ComfortReward)', lstm=True, f64=f64, kpi=kpi)
    if cfg == 'T9':
        spec = load_district(sample_schema('citylearn_challenge_2020_climate_zone_1_744h'))
        return StepWorkload(cfg, spec, E, device, rank, tuning,
                            f'citylearn_challenge_2020_climate_zone_1 (9 buildings: heat pump, electric heater, cooling + DHW tanks, battery, PV; first 744 h) '
                            f'x {E} envs per GPU, cl_step_f32 mode A (thermal district: the reference\'s full per-building energy balance), env batch sharded '
                            'over GPUs, no collective', f64=f64, kpi=kpi)
    if cfg in ('C4-B', 'C4-lean-B'):
        # BASELINE config 4 in mode B (round 5): the 1024-building district through the building-chunked fused rollout -- 24 env steps per launch,
        # unit state in registers, one cl_finish_kernel per launch for the chunks' district sums and returns
        from citylearn_amd.synthetic import tile_district
        thermal = cfg == 'C4-B'
        spec = tile_district(load_district(sample_schema('citylearn_challenge_2020_climate_zone_1_744h' if thermal else 'citylearn_challenge_2022_phase_all_720h')), 1024)
        return RolloutWorkload(cfg, spec, E, 24, device, rank, world, tuning,
                               f'synthetic 1024-building district ({"2020 climate-zone-1 device set: heat pump, heater, 2 tanks, battery" if thermal else "battery + PV"}; sizes '
                               f'jittered +-10 %) x {E} envs per GPU, cl_rollout_f32 mode B on the building-chunked district: 24 fused env steps per launch, unit state in '
                               'registers, on-device Philox4x32-10 uniform random policy, one cl_finish_kernel per launch (district sums of the last step + K-step returns); '
                               'env batch sharded over GPUs (8 x 1024 = the 8192 envs of BASELINE config 4), no collective',
                               f64=f64, valu_per_unit_step=216.6 if thermal else 100.0, valu_chain=343.1 if thermal else None,
                               valu_source=('profiles/r06w_c4b_*_sq_by_kernel.jsonl: SQ_INSTS_VALU of cl_rollout_full_kernel per launch x 64 lanes / (1024 x 1024 x 24 unit-steps) = '
                                            '216.6 lane-instructions per unit-step at two envs per lane (packed fp32), 343.1 at one under the float64 chain (the same counts at 8192 envs, '
                                            'profiles/r06z6_*, where GRBM_GUI_ACTIVE puts the clock at 2.38 GHz and SQ_ACTIVE_INST_VALU at 1.01 x the launch: frac ~ 1 = saturated; float64 instructions '
                                            'counted as one issue each)'
                                            if thermal else 'profiles/archive/r02b_rollout_pmc_by_kernel.jsonl: 100 VALU instructions per unit-step (battery + PV fused kernel)'))
    if cfg in ('C4', 'C4-lean'):
        from citylearn_amd.synthetic import tile_district
        base = 'citylearn_challenge_2020_climate_zone_1_744h' if cfg == 'C4' else 'citylearn_challenge_2022_phase_all_720h'
        spec = tile_district(load_district(sample_schema(base)), 1024)
        # district sums finished DEFERRED (cl_tuning.finish = 3): every launch folds its predecessor's chunk partial sums, cl_finish_f32 runs
        # once at the end of each captured chunk of <= 100 steps (inside the timed region); CL_TUNE_FINISH=1 restores the launch per step
        tuning = {'finish': 3, **tuning}
        return StepWorkload(cfg, spec, E, device, rank, tuning,
                            f'synthetic 1024-building district ({"2020 climate-zone-1 device set: heat pump, heater, 2 tanks, battery" if cfg == "C4" else "battery + PV"}'
                            f'; sizes jittered +-10 %) x {E} envs per GPU (8 GPUs x 1024 = the 8192 envs of BASELINE config 4), cl_step_f32 mode A, '
                            'building-chunked launch' + ('; district sums folded by the NEXT launch (deferred finish), cl_finish_f32 once per captured chunk of <= 100 steps' if tuning.get('finish') == 3 else '') + ', no collective', f64=f64, kpi=kpi)
    raise SystemExit(f'unknown --config {cfg}')


DEFAULT_ENVS = {'headline': ENVS_PER_GPU, 'C2': 4096, 'C3': 65536, 'C3-6': 65536, 'C4': 1024, 'C4-lean': 1024, 'C5': 32768, 'T9': 65536, 'C4-B': 1024, 'C4-lean-B': 1024}


# --------------------------------------------------------------------------------------------------- one rank
def dry_run_rank(args, rank: int, world: int):
    """CL_BENCH_DRY_RUN: everything but the GPU -- rendezvous (gloo), barrier, MAX-over-ranks, per-rank gather, one JSON line."""
    from citylearn_amd.parallel import gather_seconds, init_control_plane, reduce_max_seconds
    dist = init_control_plane(rank, world, None, 'gloo') if world > 1 else None
    mine = 1e-3 * (rank + 1) * args.steps                       # a different "wall time" per rank: the line must carry the MAX
    wall = reduce_max_seconds(mine, dist, 'cpu')
    per_rank = gather_seconds(mine, dist, 'cpu')
    if rank == 0:
        print(json.dumps({'metric': METRIC, 'value': world * 17 * DEFAULT_ENVS[args.config] * args.steps / wall, 'unit': 'building-timesteps/s',
                          'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': wall / args.steps * 1e3,
                          'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
                          'config': {'workload': 'DRY RUN (CL_BENCH_DRY_RUN): no GPU work, synthetic timings'}, 'dry_run': True,
                          'world_size_seen': world if dist is None else dist.get_world_size(), 'rccl_world_size': None, 'control_backend': 'gloo' if dist is not None else None,
                          'rank_ms_per_step': [s / args.steps * 1e3 for s in per_rank]}))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def run_rank(args):
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    if rank != 0:
        # stdout belongs to rank 0's one JSON line: an external launcher (torch.distributed.run) merges every rank's stdout into its own,
        # and libraries print there (RCCL's banner).  The other ranks' descriptor 1 goes to stderr for the life of the process.
        sys.stdout.flush()
        os.dup2(2, 1)
    if os.environ.get('CL_BENCH_DRY_RUN'):
        import torch
        if torch.cuda.is_available():
            # the dry run prints a complete metric line of made-up timings: it must be impossible to get one from a box that could measure
            raise SystemExit('CL_BENCH_DRY_RUN is a CPU-only launcher test hook: refused because a GPU is visible (unset it to measure)')
        return dry_run_rank(args, rank, world)

    import torch
    from citylearn_amd.parallel import gather_seconds, init_control_plane, reduce_max_seconds
    n_dev = torch.cuda.device_count()
    if n_dev == 0:
        raise SystemExit('bench.py needs a GPU (no HIP device visible); there is no CPU path')
    oversubscribed = world > n_dev
    if oversubscribed and not os.environ.get('CL_BENCH_OVERSUBSCRIBE'):
        raise SystemExit(f'--gpus {world} but only {n_dev} device(s) visible (set CL_BENCH_OVERSUBSCRIBE=1 to let ranks share a GPU: a plumbing test, '
                         'not a scaling measurement)')
    dev_index = local_rank % n_dev
    torch.cuda.set_device(dev_index)
    device = f'cuda:{dev_index}'
    # N > 1: this rank's launch thread on cores of the NUMA node its GPU hangs off (CL_BENCH_NO_PIN=1 leaves the affinity mask alone)
    from citylearn_amd.parallel import pin_rank_to_gpu_node
    affinity = None
    if world > 1 and not os.environ.get('CL_BENCH_NO_PIN'):
        affinity = pin_rank_to_gpu_node(dev_index, local_rank, int(os.environ.get('LOCAL_WORLD_SIZE', world)))
    dist, backend = None, None
    if world > 1 or os.environ.get('CL_BENCH_FORCE_DIST'):
        backend = os.environ.get('CL_BENCH_CONTROL') or ('gloo' if oversubscribed else 'nccl')        # RCCL refuses two ranks on one device
        dist = init_control_plane(rank, world, device, backend)
        backend = dist.control_backend                        # 'gloo' also when RCCL could not come up (the line then carries `control_fallback`)
    ctl_device = device if backend == 'nccl' else 'cpu'

    tuning = {k[len('CL_TUNE_'):].lower(): int(v) for k, v in os.environ.items() if k.startswith('CL_TUNE_')}   # e.g. CL_TUNE_ENVMAJOR=2
    use_graph = not args.no_graph
    cfg = args.config
    E = args.envs_per_gpu or DEFAULT_ENVS[cfg]

    def measure(wl, warmup: int, steps: int, reps: int, kernel_steps: int):
        runner = Runner(wl.step_fn, wl.period, torch.cuda.Stream(device=device), use_graph, getattr(wl, 'flush', None))
        rep, kernel_s = timed_reps(runner, wl.reset, warmup, steps, reps, dist, kernel_steps)
        walls = [reduce_max_seconds(w, dist, ctl_device) for w, _ in rep]         # MAX over ranks per repetition
        evs = [reduce_max_seconds(e, dist, ctl_device) for _, e in rep]
        mine = statistics.median(w for w, _ in rep)
        return walls, evs, reduce_max_seconds(kernel_s, dist, ctl_device), gather_seconds(mine, dist, ctl_device), gather_seconds(kernel_s, dist, ctl_device)

    f64 = PRECISIONS[args.precision]                             # StepEngine's f64_maps: 'chain' (default) | False | True
    default_precision = args.precision == 'chain'
    wl = build_workload(cfg, E, device, rank, world, tuning, f64, args.kpi, args.table_hours)
    heavy = cfg in ('C3', 'C3-6', 'C5', 'C4-B', 'C4-lean-B')     # ~100 us .. 1 ms per step: fewer steps in the kernel-time bracket
    walls, evs, launch_s, per_rank, per_rank_kernel = measure(wl, args.warmup, args.steps, args.reps, max(args.steps, 200 if heavy else 2000))
    wall_med = statistics.median(walls)
    roof = wl.roofline(launch_s)
    roof['launch_us_how'] = ('HIP events on the launch stream around max(K, 2000) consecutive steps (pre-replayed 100-step hipGraphs) enqueued behind a '
                             'lead-in chunk: kernel time only') if not heavy else 'HIP events on the launch stream around max(K, 200) consecutive steps behind a lead-in chunk'
    roof['timed_region_event_us_per_step'] = [e / args.steps * 1e6 for e in evs]
    if roof['bound'] == 'hbm':
        pattern = {'headline': 'r*_bench_pmc_summary.json', 'C2': 'r*_c2_pmc_summary.json', 'C4': 'r*_c4_pmc_summary.json',
                   'C4-lean': 'r*_c4lean_pmc_summary.json', 'T9': 'r*_kpi_t9_pmc_summary.json' if args.kpi else 'r*_t9_pmc_summary.json'}.get(cfg, 'none')
        roof['traffic'], roof['traffic_source'] = _pmc_traffic(pattern, wl.kernels or '') if E == DEFAULT_ENVS[cfg] else (None, None)
        if args.traffic_summary:
            f = Path(args.traffic_summary)
            c = json.loads(f.read_text())
            if 'FETCH_SIZE' in c and 'WRITE_SIZE' in c and any(k and k in c.get('_kernel', {}).get('kernel', '') for k in (wl.kernels or '').split('+')):
                roof['traffic'], roof['traffic_source'] = (2.0 * c['FETCH_SIZE']['mean'] + c['WRITE_SIZE']['mean']) * 1024.0, f.name
    prec_args = ['--precision', args.precision]
    if cfg == 'headline' and world == 1 and rank == 0 and not args.no_traffic_pass and E == DEFAULT_ENVS[cfg] and not args.kpi:
        live, how = _live_traffic(wl.kernels or '', ['--table-hours', str(args.table_hours), *prec_args])
        if live is not None:
            roof['traffic_committed_file'] = {'traffic': roof.get('traffic'), 'source': roof.get('traffic_source')}
            roof['traffic'], roof['traffic_source'] = live, how
        else:
            roof['traffic_live_error'] = how
    if cfg == 'headline':
        # (`bound` keeps the contract's vocabulary -- this path has no MFMA, so "hbm" -- but at THIS shape the bytes come out of the Infinity Cache)
        roof['residency'] = 'infinity-cache (fabric bandwidth, not HBM)' if E * 17 * 52 < 256e6 else 'hbm'
    units_per_step, n_bldg, spec, tables, what, env_pitch = wl.units_per_step, wl.eng.n_bldg, wl.spec, wl.tables, wl.what, wl.eng.env_pitch

    def streaming_entry(f64_s, with_traffic: bool):
        """The same tables x STREAMING_ENVS envs: 659 MB of algorithmic traffic per launch, far beyond the 256 MB Infinity Cache -- the HBM-true figure."""
        s_steps = 20
        wl_s = build_workload(cfg, STREAMING_ENVS, device, rank, world, tuning, f64_s, args.kpi, args.table_hours)
        _, _, launch, _, _ = measure(wl_s, 5, s_steps, 3, 2000)
        a = wl_s.units_per_step * wl_s.bytes_per_unit() / launch / 1e9
        s_units, s_bpu, s_kernels, s_pitch = wl_s.units_per_step, wl_s.bytes_per_unit(), wl_s.kernels, wl_s.eng.env_pitch
        s_traffic, s_source = _pmc_traffic('r*_streaming_pmc_summary.json', wl_s.kernels or '')
        s_live_error = None
        wl_s = None                                        # (the child runs allocate the same 17 x 1 048 576 planes)
        torch.cuda.empty_cache()
        if with_traffic and world == 1 and rank == 0 and not args.no_traffic_pass and not args.kpi:
            live, how = _live_traffic(s_kernels or '', ['--envs-per-gpu', str(STREAMING_ENVS), '--table-hours', str(args.table_hours),
                                                        '--precision', {'chain': 'chain', False: 'fp32', True: 'f64'}[f64_s]], steps=30, warmup=5)
            if live is not None:
                s_traffic, s_source = live, how
            else:
                s_live_error = how
        return {'workload': f'same tables x {STREAMING_ENVS} envs per GPU ({s_units * s_bpu / 1e6:.0f} MB of algorithmic traffic per launch, beyond the 256 MB Infinity Cache)',
                'bound': 'hbm', 'achieved': a, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': a / HBM_PEAK_GBS,
                'frac_vs_measured_copy': a / HBM_MEASURED_COPY_GBS, 'kernel': s_kernels, 'launch_us': launch * 1e6,
                'algorithmic_bytes_per_unit': s_bpu, 'units_per_launch': s_units, 'steps': s_steps, 'env_pitch': s_pitch, 'traffic': s_traffic, 'traffic_source': s_source,
                **({'traffic_live_error': s_live_error} if s_live_error else {}),
                'value': world * s_units / launch}

    if cfg == 'headline' and not args.no_streaming and E == ENVS_PER_GPU:
        # `roofline` of the line = the HBM-TRUE figure (the field the metric calls "HBM GB/s vs roofline" must be an HBM
        # number): the step kernel on the same tables at 1 048 576 envs, measured live in this run like the metric shape -- HIP events on the launch
        # stream, counters from rocprofv3 child passes.  The launch the VALUE is timed on (17 x 65 536: its 58 MB working set sits in the
        # Infinity Cache, so its bytes / time is fabric bandwidth) keeps its full entry as `roofline.metric_shape`.
        wl = None
        torch.cuda.empty_cache()
        stream = streaming_entry(f64, True)
        metric_shape = roof
        metric_shape['note'] = ('the launch `value` / `ms_per_step` are timed on: working set (state 13 MB + outputs 9 MB + action ring 36 MB) fits the 256 MB Infinity '
                                'Cache, so achieved / frac here are FABRIC bandwidth against the HBM peak -- not the HBM fraction (that is the parent object)')
        roof = {k: stream[k] for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'frac_vs_measured_copy', 'kernel', 'launch_us', 'algorithmic_bytes_per_unit',
                                       'units_per_launch', 'traffic', 'traffic_source', 'env_pitch')}
        roof.update({'shape': stream['workload'], 'steps': stream['steps'], 'value_at_this_shape': stream['value'],
                     **({'traffic_live_error': stream['traffic_live_error']} if 'traffic_live_error' in stream else {}),
                     'launch_us_how': 'HIP events on the launch stream around 2000 consecutive steps behind a lead-in chunk (pre-replayed hipGraphs): kernel time only',
                     'note': 'HBM-true roofline of the dominant kernel: the metric\'s 17 x 65 536 launch is cache-resident (see metric_shape), so the fraction of the HBM '
                             'roofline is measured where the working set is 2.6 x the Infinity Cache -- same tables, same precision model, same run',
                     'metric_shape': metric_shape})

    if cfg == 'headline' and default_precision and not args.kpi and not args.no_side_entries:
        # the all-fp32 battery map (`StepEngine(f64_maps=False)`): the throughput mode a user can opt into -- 1e-4 teacher-forced, but free-running it
        # drifts to 6.9 x the bar over config 1's year (tests/test_gpu_parity.py::test_full_year_free_running_every_step), which is why the
        # line is not quoted on it.  Its numbers at both shapes, same run:
        wl = None
        torch.cuda.empty_cache()
        wl_f = build_workload(cfg, E, device, rank, world, tuning, False, False, args.table_hours)
        _, _, f_launch, _, _ = measure(wl_f, 50, 200, 1, 2000)
        rf = wl_f.roofline(f_launch)
        m_us = (roof.get('metric_shape') or roof)['launch_us']
        side = {'what': 'StepEngine(f64_maps=False): all-fp32 battery map; same planes, same bytes; NOT the default (free-running drift beyond 1e-4)',
                'metric_shape': {'kernel': rf['kernel'], 'launch_us': rf['launch_us'], 'frac_of_hbm_peak': rf['frac'], 'achieved': rf['achieved'], 'unit': rf['unit'],
                                 'value': world * wl_f.units_per_step / f_launch, 'speedup_vs_default': m_us / rf['launch_us']}}
        wl_f = None
        torch.cuda.empty_cache()
        if 'metric_shape' in roof:
            fs = streaming_entry(False, False)
            side['hbm_streaming'] = {k: fs[k] for k in ('kernel', 'launch_us', 'frac', 'achieved', 'unit', 'value')}
            side['hbm_streaming']['speedup_vs_default'] = roof['launch_us'] / fs['launch_us']
        roof['fp32_map'] = side

    if cfg in ('C2', 'T9') and not args.no_side_entries and not args.kpi:
        # BASELINE config 2 (17 x 4 096 envs) is 2.7 MB per step: one launch per step is launch latency whatever the kernel does.
        # What a user who wants THROUGHPUT at this batch size gets is mode B -- 24 fused steps per launch, state in registers -- measured here beside mode A.
        wl = None
        torch.cuda.empty_cache()
        # (T9, round 6: the thermal district through the packed unit of cl_rollout_full_kernel)
        if cfg == 'C2':
            spec_b = load_c2_spec(args.table_hours)
        else:
            from citylearn_amd import load_district
            from citylearn_amd.data import sample_schema
            spec_b = load_district(sample_schema('citylearn_challenge_2020_climate_zone_1_744h'))
        wl_b = RolloutWorkload(cfg + '-B', spec_b, E, 24, device, rank, world, tuning, f'mode B at the {cfg} shape', f64=f64)
        _, _, b_launch, _, _ = measure(wl_b, 10, 100, 1, 200)
        roof['mode_b'] = {'what': f'cl_rollout_f32 at the same shape ({len(spec_b.buildings)} buildings x {E} envs): 24 fused env steps per launch, on-device Philox policy, state in registers',
                          'kernel': wl_b.kernels, 'launch_us_per_24_steps': b_launch * 1e6, 'us_per_step': b_launch * 1e6 / 24,
                          'value': world * wl_b.units_per_step / b_launch, 'speedup_vs_mode_a': roof['launch_us'] / (b_launch * 1e6 / 24)}
        wl_b = None
        torch.cuda.empty_cache()

    # N > 1 on a real node: BASELINE configs 4 and 5 measured in the same lease (their per-GPU shards, weak scaling like the headline) -- the
    # driver's scaling run is the only time anybody sees N > 1, so the line carries them as `extra_configs` (--no-extra-configs skips them)
    extra = {}
    want_extra = (world > 1 and not oversubscribed and not args.no_extra_configs) or os.environ.get('CL_BENCH_EXTRA_CONFIGS') == '1'      # (the hook: 1-GPU tests)
    if cfg == 'headline' and want_extra:
        wl = None
        torch.cuda.empty_cache()
        # `fixed-65536`: north_star's sentence also reads as ONE 65 536-env batch split over the N GPUs (strong scaling: 65 536 / N envs per rank --
        # at N = 8 a launch-bound 8 192-env kernel); the headline above is the weak-scaling reading (65 536 envs PER GPU).  Both are in the line.
        # (round 6: BASELINE config 4 also in mode B -- `C4-B` / `C4-lean-B`, one step = one fused 24-step launch -- the config's fast path since the
        #  thermal district runs the packed unit there)
        for xc in ('fixed-65536', 'C4', 'C4-lean', 'C5', 'C4-B', 'C4-lean-B'):
            x_steps, x_warm = {'C5': (200, 30), 'C4-B': (100, 10), 'C4-lean-B': (200, 20)}.get(xc, (2000, 200))
            if os.environ.get('CL_BENCH_EXTRA_CONFIGS') == '1':
                x_steps, x_warm = x_steps // 10, x_warm // 10
            if xc == 'fixed-65536':
                e_fixed = max(4, (ENVS_PER_GPU // world) // 4 * 4)
                wl_x = build_workload('headline', e_fixed, device, rank, world, tuning, f64, False, args.table_hours)
                wl_x.what += f'; FIXED total batch: {ENVS_PER_GPU} envs split over {world} rank(s) = {e_fixed} envs per GPU (strong scaling)'
            else:
                wl_x = build_workload(xc, DEFAULT_ENVS[xc], device, rank, world, tuning, f64, False, args.table_hours)
            x_walls, _, x_launch, x_rank, x_rank_k = measure(wl_x, x_warm, x_steps, 3, x_steps if xc in ('C5', 'C4-B', 'C4-lean-B') else 2000)
            x_wall = statistics.median(x_walls)
            extra[xc] = {'workload': wl_x.what, 'value': world * wl_x.units_per_step * x_steps / x_wall, 'unit': 'building-timesteps/s',
                         'ms_per_step': x_wall / x_steps * 1e3, 'steps': x_steps, 'warmup': x_warm, 'reps': 3, 'envs_per_gpu': wl_x.E,
                         'scaling': 'strong' if xc == 'fixed-65536' else 'weak',
                         'rank_ms_per_step': [v / x_steps * 1e3 for v in x_rank], 'rank_launch_us': [k * 1e6 for k in x_rank_k],
                         'roofline': wl_x.roofline(x_launch)}
            del wl_x
            torch.cuda.empty_cache()
    affinities = None
    if dist is not None and world > 1:
        affinities = [None] * world
        torch.distributed.all_gather_object(affinities, affinity)            # (default group = gloo: host objects)
    if rank == 0:
        n_distinct = min(world, n_dev)
        out = {
            'metric': METRIC,
            'value': world * units_per_step * args.steps / wall_med,
            'unit': 'building-timesteps/s',
            'n_gpus': world if not oversubscribed else n_distinct, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': wall_med / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': {'chain': 'f32 (battery soc chain in f64)', 'fp32': 'f32', 'f64': 'f32 (battery map in f64)'}[args.precision], 'data': 'synthetic',
            'config': {'workload': what, 'name': cfg, 'envs_per_gpu': E, 'buildings': n_bldg, 'env_pitch': env_pitch,
                       'precision': {'chain': 'CLD_F64_CHAIN (StepEngine default): 1e-4 free-running over whole episodes', 'fp32': 'all-fp32 battery map (f64_maps=False)',
                                     'f64': 'CLD_F64_MAPS: bit-identical battery state'}[args.precision],
                       'launch': 'hipGraph replay' if use_graph else 'eager', 'reward': 'ComfortReward' if cfg in ('C3', 'C3-6') else 'RewardFunction',
                       'reps': args.reps, 'statistic': 'median of reps (each: MAX over ranks)',
                       **({'k_steps_per_launch': 24, 'step': 'one fused 24-step launch'} if cfg in ('C5', 'C4-B', 'C4-lean-B') else {})},
            'ranks': world, 'world_size_seen': world if dist is None else dist.get_world_size(),
            # ranks inside the RCCL communicator itself (`world_size_seen` is the gloo group every rank joins first); null when the control plane is gloo
            'rccl_world_size': None if dist is None else dist.rccl_world_size, 'control_backend': backend, **({'control_fallback': dist.control_fallback} if dist is not None and dist.control_fallback else {}),
            'rank_ms_per_step': [s / args.steps * 1e3 for s in per_rank],
            # kernel time only, per rank (HIP events around back-to-back steps): next to rank_ms_per_step it separates what the GPU took from
            # what the host's graph submission added -- 8 ranks share the host's usable cores
            'rank_launch_us': [k * 1e6 for k in per_rank_kernel],
            'rep_ms_per_step': [w / args.steps * 1e3 for w in walls],
            'roofline': roof,
        }
        if extra:
            out['extra_configs'] = extra
        if affinities is not None:
            out['rank_affinity'] = affinities
        if oversubscribed:
            out['oversubscribed'] = True
            out['config']['note'] = (f'{world} ranks share {n_dev} GPU(s) (CL_BENCH_OVERSUBSCRIBE): exercises the multi-rank plumbing, NOT a scaling measurement; '
                                     'n_gpus = distinct devices')
        if cfg == 'headline' and world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(spec, tables)
            out['dropin'] = dropin_timing(device, out['cpu_baseline'])
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


# --------------------------------------------------------------------------------------------------- entry
def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=None, help='timed steps K (default: 5000; 300 for C3, 200 for C5)')
    ap.add_argument('--warmup', type=int, default=None, help='untimed warmup steps W (default: 300; 30 for C3 / C5)')
    ap.add_argument('--reps', type=int, default=5, help='timed repetitions of the K steps; the median is reported')
    ap.add_argument('--config', choices=CONFIGS, default='headline')
    ap.add_argument('--envs-per-gpu', type=int, default=None)
    ap.add_argument('--table-hours', type=int, choices=(720, 8760), default=8760,
                    help='2022_phase_all configs (headline, C2, C5): table rows = the whole year (default) or the 720-hour cut')
    ap.add_argument('--traffic-summary', default=None, help='a scripts/pmc_summary.py file to take roofline.traffic from (profile runs at sizes other than '
                                                            'the default; used only if it was collected on the kernel this run launches)')
    ap.add_argument('--no-graph', action='store_true', help='launch every step from Python instead of hipGraph replay')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extra-configs', action='store_true', help='N > 1, headline: skip the C4 / C4-lean / C5 lines measured in the same run (`extra_configs`)')
    ap.add_argument('--no-traffic-pass', action='store_true', help='headline, N = 1: skip the two rocprofv3 --pmc child runs that measure roofline.traffic live')
    ap.add_argument('--no-streaming', action='store_true', help='skip the 17 x 1 048 576 HBM-streaming roofline entry of the headline')
    ap.add_argument('--no-side-entries', '--no-chain-entry', dest='no_side_entries', action='store_true',
                    help='skip the all-fp32 side entry of the headline line (roofline.fp32_map)')
    ap.add_argument('--precision', choices=tuple(PRECISIONS), default='chain',
                    help="battery-map precision model: 'chain' (CLD_F64_CHAIN, the engine's default: 1e-4 free-running), 'fp32' (all-fp32 map: the opt-in "
                         "throughput mode), 'f64' (CLD_F64_MAPS: the reference's mixed precision, bit-identical battery state)")
    ap.add_argument('--fp32-map', action='store_true', help='= --precision fp32')
    ap.add_argument('--f64-maps', action='store_true', help='= --precision f64')
    ap.add_argument('--f64-chain', action='store_true', help='= --precision chain (the default since round 6)')
    ap.add_argument('--kpi', action='store_true', help='CLD_KPI: update the streaming KPI accumulators every step (mode A-kpi of SURVEY 8d; step configs)')
    ap.add_argument('--launch-timeout', type=float, default=None, help='seconds after which self-spawned ranks are terminated')
    args = ap.parse_args(argv)
    if args.fp32_map:
        args.precision = 'fp32'
    elif args.f64_maps:
        args.precision = 'f64'
    elif args.f64_chain:
        args.precision = 'chain'
    heavy = args.config in ('C3', 'C3-6', 'C5', 'C4-B', 'C4-lean-B')
    if args.steps is None:
        args.steps = {'C3': 300, 'C3-6': 200, 'C5': 200, 'C4-B': 100, 'C4-lean-B': 200}.get(args.config, 5000)
    if args.warmup is None:
        args.warmup = 30 if heavy else 300
    return args


def main(argv=None):
    args = parse_args(argv)
    if args.gpus > 1 and 'RANK' not in os.environ and 'WORLD_SIZE' not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves, one process per GPU
        from citylearn_amd.parallel import launch_ranks
        rc, out0 = launch_ranks([sys.executable, str(Path(__file__).resolve())] + list(sys.argv[1:] if argv is None else argv), args.gpus,
                                timeout=args.launch_timeout)
        lines = [ln for ln in out0.splitlines() if ln.startswith('{')]
        for ln in out0.splitlines():
            if not ln.startswith('{'):
                print(ln, file=sys.stderr)              # anything else rank 0 wrote to stdout: keep stdout to the one JSON line
        if lines:
            print(lines[-1])
        elif rc == 0:
            rc = 1
            print('bench.py: rank 0 printed no JSON line', file=sys.stderr)
        raise SystemExit(rc)
    run_rank(args)


if __name__ == '__main__':
    main()
