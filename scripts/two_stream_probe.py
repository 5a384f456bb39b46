"""Do independent env batches on separate HIP streams fill each other's end-of-kernel drain?  N engines (each its own state, actions,
outputs, stream, K-step hipGraph) replayed concurrently vs one engine over the whole batch.  GPU box; output tracked under profiles/."""
import sys, time
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests'))
from golden_util import golden
from citylearn_amd.engine import StepEngine

STEPS, REPS = 100, 20


def build(tab, spec, E, n_streams, tuning=None):
    low, high = spec.action_limits()
    lo, hi = torch.from_numpy(low).cuda(), torch.from_numpy(high).cuda()
    parts = []
    for i in range(n_streams):
        eng = StepEngine(tab, E // n_streams, tuning=tuning, env_offset=i * (E // n_streams))
        eng.trace_kernels()
        acts = [lo[:, None] + torch.rand((len(low), eng.n_env), device='cuda') * (hi - lo)[:, None] for _ in range(4)]
        stream = torch.cuda.Stream()
        with torch.cuda.stream(stream):
            for t in range(3):
                eng.step(acts[t % 4], 1 + t)
            stream.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=stream):
                for t in range(STEPS):
                    eng.step(acts[t % 4], 1 + t % 600)
            g.replay(); stream.synchronize()
        parts.append((eng, acts, stream, g))
    return parts


def run(parts):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(REPS):
        for eng, acts, stream, g in parts:
            with torch.cuda.stream(stream):
                g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (STEPS * REPS) * 1e6


for name, E in (('g2022_all', 65536), ('g2022_all', 131072), ('g2022_all', 262144), ('g2020_cz1', 65536), ('g2022_all', 1048576)):
    spec = golden(name).spec(); tab = spec.episode_tables(0)
    for n in (1, 2, 4):
        parts = build(tab, spec, E, n)
        us = min(run(parts) for _ in range(3))
        eng = parts[0][0]
        bpu = eng.algorithmic_bytes_per_unit()
        units = E * eng.n_bldg
        print(f'{name} {eng.n_bldg} x {E} as {n} x {E // n} on {n} stream(s) [{eng.last_kernels}]: {us:.2f} us per whole-batch step, '
              f'{units / us * 1e6:.3e} building-timesteps/s, {units * bpu / us / 1e3:.0f} GB/s = {units * bpu / us / 1e3 / 8000:.3f}', flush=True)
        del parts
        torch.cuda.empty_cache()
