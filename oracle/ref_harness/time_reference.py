"""Time the REFERENCE's own step path -- `CityLearnEnv.step` (/root/reference/citylearn/citylearn.py:978-1056) -- on this host.

TEST / MEASUREMENT INFRASTRUCTURE (oracle side): imports the reference through `ref_env` (gymnasium / simplejson stand-ins,
seeded cache).  `/root/reference` exists only in the build container; `stage_reference.py` (run by
`__graft_entry__.build()`) stages the package + the two datasets this timing needs into the git-ignored `oracle/_ref/reference/`,
which travels to the GPU box with the snapshot.  `bench.py`'s `cpu_baseline` leg runs this script there (`--root oracle/_ref/reference
--c1-budget 60 --out -`, a bounded sample) and reports the result as `cpu_baseline` (kind "reference"), host = the GPU box, the C port beside it.
The long form (C1 exactly + 1000 steps on every core) is committed as `profiles/reference_cpu_timing.json`.

    python oracle/ref_harness/time_reference.py [--root DIR] [--steps 1000] [--procs N] [--skip-c1] [--out FILE|-]

Two measurements (SURVEY.md 8d "CPU baseline beside it"):
  (i)  C1 exactly: citylearn_challenge_2022_phase_1 (5 buildings), ONE process, the full 8759-step episode, actions
       np.random.RandomState(0).uniform(-1, 1) per building per step;
  (ii) N independent processes (default: every core of this host) each stepping its own citylearn_challenge_2022_phase_all env
       (17 buildings) for `--steps` steps -- the path has no intra-step threading, so processes are the only CPU parallelism.
Unit: building-timesteps per second (one (env, building) advanced by one step), per core and aggregate.
"""
import argparse
import json
import multiprocessing as mp
import os
import platform
import sys
import time
from pathlib import Path

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))


def _episode(dataset: str, steps, seed: int, barrier=None, budget_s=None, fallback_steps=2000, probe_steps=500):
    """`budget_s` (C1 on the bench host): after `probe_steps` steps the full episode's duration is projected from the rate so far; if it
    exceeds the budget the episode stops at `fallback_steps` (the count actually run is what the caller reports)."""
    import numpy as np
    import ref_env
    ref_env.setup_reference()
    from citylearn.citylearn import CityLearnEnv
    env = CityLearnEnv(ref_env.dataset_schema(dataset))
    env.reset()
    B = len(env.buildings)
    n = env.time_steps - 1 if steps is None else min(int(steps), env.time_steps - 1)
    rng = np.random.RandomState(seed)
    acts = [[list(rng.uniform(-1, 1, size=len(space.low))) for space in env.action_space] for _ in range(n)]
    if barrier is not None:
        barrier.wait()
    t0 = time.perf_counter()
    done = 0
    for a in acts:
        env.step(a)
        done += 1
        if budget_s is not None and done == probe_steps and (time.perf_counter() - t0) / done * n > budget_s:
            n = min(n, max(int(fallback_steps), probe_steps))
        if done >= n:
            break
    dt = time.perf_counter() - t0
    return B, done, dt


def _worker(dataset, steps, seed, barrier, q):
    q.put(_episode(dataset, steps, seed, barrier))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=1000)
    ap.add_argument('--procs', type=int, default=os.cpu_count() or 1)
    ap.add_argument('--out', default=str(HERE.parent.parent / 'profiles' / 'reference_cpu_timing.json'),
                    help="file, or '-' = one JSON line on stdout (what bench.py reads)")
    ap.add_argument('--root', default=None, help='reference tree to import (default: $CITYLEARN_REFERENCE_ROOT, /root/reference, '
                                                 'else the staging under oracle/_ref/reference)')
    ap.add_argument('--skip-c1', action='store_true', help='skip the single-process full C1 episode (about 2 minutes)')
    ap.add_argument('--c1-budget', type=float, default=None, help='seconds the C1 episode may take: if the first 500 steps project the full 8759 '
                                                                  'beyond it, only the first --c1-fallback-steps are run (and the count is stated)')
    ap.add_argument('--c1-fallback-steps', type=int, default=2000)
    args = ap.parse_args()
    if args.root:
        os.environ['CITYLEARN_REFERENCE_ROOT'] = str(Path(args.root).resolve())       # inherited by the spawned workers
    import ref_env
    log = sys.stderr if args.out == '-' else sys.stdout

    c1 = None
    if not args.skip_c1:
        B, n, dt = _episode('citylearn_challenge_2022_phase_1', None, 0, budget_s=args.c1_budget, fallback_steps=args.c1_fallback_steps)
        c1 = {'dataset': 'citylearn_challenge_2022_phase_1', 'buildings': B, 'steps': n, 'seconds': round(dt, 3), 'processes': 1,
              'building_timesteps_per_s': B * n / dt,
              'episode': 'full (8759 transitions)' if n >= 8759 else f'first {n} steps (the full episode projected beyond the {args.c1_budget:.0f} s budget)'}
        print('C1:', c1, file=log, flush=True)

    ctx = mp.get_context('spawn')
    barrier, q = ctx.Barrier(args.procs), ctx.Queue()
    procs = [ctx.Process(target=_worker, args=('citylearn_challenge_2022_phase_all', args.steps, 100 + r, barrier, q)) for r in range(args.procs)]
    for p in procs:
        p.start()
    res = [q.get() for _ in procs]
    for p in procs:
        p.join()
    slowest = max(r[2] for r in res)
    units = sum(r[0] * r[1] for r in res)
    par = {'dataset': 'citylearn_challenge_2022_phase_all', 'buildings': res[0][0], 'steps': res[0][1], 'processes': args.procs,
           'seconds_slowest_process': round(slowest, 3), 'building_timesteps_per_s': units / slowest,
           'building_timesteps_per_s_per_core': units / slowest / args.procs}
    print('parallel:', par, file=log, flush=True)

    cpu = platform.processor() or platform.machine()
    try:
        cpu = next(line.split(':', 1)[1].strip() for line in open('/proc/cpuinfo') if line.startswith('model name'))
    except Exception:
        pass
    out = {'kind': 'reference', 'what': 'CityLearnEnv.step of the reference (citylearn.py:978-1056, v2.4.2), numpy / Python',
           'value': par['building_timesteps_per_s'], 'unit': 'building-timesteps/s', 'cores': args.procs,
           'host': f'{platform.node()} ({cpu}, {os.cpu_count()} logical cores)', 'reference_root': str(ref_env.REFERENCE_ROOT),
           'date': time.strftime('%Y-%m-%d'), 'sample': f'{args.procs} processes x 17 buildings x {par["steps"]} steps', 'c1_single_process': c1,
           'all_cores': par, 'script': 'oracle/ref_harness/time_reference.py'}
    if args.out == '-':
        print(json.dumps(out), flush=True)
    else:
        Path(args.out).write_text(json.dumps(out, indent=1) + '\n')
        print('wrote', args.out)


if __name__ == '__main__':
    main()
