"""How to enqueue a SHORT timed region (the driver's `--steps 20`): one pre-captured hipGraph of K steps vs K launches from one C call
(`StepEngine.step_many` = cl_rollout_seq_f32 with an open-loop action tensor).  Wall time per step, synchronize to synchronize, like
bench.py's timed region (GPU box)."""
import statistics
import sys
import time
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
acts = (torch.rand((8, eng.n_act_cols, E), device='cuda') * 2 - 1).contiguous()
stream = torch.cuda.Stream()
for K in (20, 50, 100):
    with torch.cuda.stream(stream):
        for i in range(K):
            eng.step(acts[i % 8], 1 + i)
        stream.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            for i in range(K):
                eng.step(acts[i % 8], 1 + i)
        g.replay(); stream.synchronize()
        rows = {}
        for label in ('graph', 'step_many x8', 'graph', 'step_many x8'):
            walls = []
            for _ in range(15):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                if label == 'graph':
                    g.replay()
                else:
                    i = 0
                    while i < K:
                        n = min(8, K - i)
                        eng.step_many(acts[:n], 1 + i)
                        i += n
                stream.synchronize()
                torch.cuda.synchronize()
                walls.append((time.perf_counter() - t0) / K * 1e6)
            rows.setdefault(label, []).append(statistics.median(walls))
    print(f'K={K}: ' + ' | '.join(f'{k}: ' + ' / '.join(f'{v:.2f}' for v in vs) for k, vs in rows.items()) + ' us per step (median of 15 regions, two rounds)', flush=True)
