"""Round 5: does the HBM-streaming shape (17 x 1 048 576) speed up under sustained load?  One engine, one captured 100-step graph, replayed in
blocks of 1 000 steps for ~30 s; per block the average step time by HIP events.  GPU box."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
from citylearn_amd import load_district
from citylearn_amd.data import sample_schema
from citylearn_amd.engine import StepEngine
tab = load_district(sample_schema()).episode_tables(0)
for E in [int(x) for x in sys.argv[1:]] or (1 << 20, 65536):
    eng = StepEngine(tab, E)
    acts = [(torch.rand((eng.n_act_cols, E), device='cuda') * 2 - 1) for _ in range(2)]
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        for t in range(3): eng.step(acts[t % 2], 1 + t)
        stream.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            for t in range(100): eng.step(acts[t % 2], 1 + t % 600)
        g.replay(); stream.synchronize()
        t0 = time.perf_counter()
        out = []
        reps = 10 if E >= (1 << 19) else 100
        while time.perf_counter() - t0 < (25 if E >= (1 << 19) else 10):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(reps): g.replay()
            e1.record(stream); stream.synchronize()
            out.append((time.perf_counter() - t0, e0.elapsed_time(e1) / (100 * reps) * 1e3))
    print(f'E={E}: ' + ' '.join(f'{t:.1f}s:{us:.2f}' for t, us in out[::max(1, len(out) // 40)]), flush=True)
    del eng, acts, g; torch.cuda.empty_cache()
