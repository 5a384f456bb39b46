"""The driver times bench.py with --steps 20: what does a 20-step timed region cost per step as one hipGraph replay, as 20 eager
Python calls, and as one C call that enqueues the 20 launches (StepEngine.step_many -> cl_rollout_seq_f32)?  Wall clock between
synchronisations, like bench.py."""
import sys, time, statistics
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from citylearn_amd import load_district
from citylearn_amd.data import sample_schema
from citylearn_amd.engine import StepEngine

spec = load_district(sample_schema('citylearn_challenge_2022_phase_all_720h'))
tab = spec.episode_tables(0)
E = 65536
eng = StepEngine(tab, E)
for K in (20, 100):
    ring = torch.rand((K, eng.n_act_cols, E), device='cuda') * 2 - 1
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        for k in range(5):
            eng.step(ring[k], k)
        stream.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            for k in range(K):
                eng.step(ring[k], 5 + k)
        g.replay(); stream.synchronize()

        def timed(fn, reps=9):
            out = []
            for _ in range(reps):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                fn()
                stream.synchronize(); torch.cuda.synchronize()
                out.append((time.perf_counter() - t0) / K * 1e6)
            return statistics.median(out), min(out)

        def eager():
            for k in range(K):
                eng.step(ring[k], 5 + k)
        print(f'K = {K}: graph replay {timed(g.replay)}  eager python {timed(eager)}  one C call {timed(lambda: eng.step_many(ring, 5))} us per step (median, min)', flush=True)
