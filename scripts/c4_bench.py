"""BASELINE configs 3 / 4 timing on the GPU box (full kernel): 2023 schema x 65536 envs, 2020 schema x 65536 envs,
synthetic 1024-building district x 1024 envs per GPU (config 4's per-GPU shard)."""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests'))
from golden_util import golden
from citylearn_amd.engine import StepEngine
from citylearn_amd.synthetic import tile_district

def measure(eng, acts, steps=60, reps=5):
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        for t in range(3): eng.step(acts[t % 2], 1 + t)
        stream.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            for t in range(steps): eng.step(acts[t % 2], 1 + t % 600)
        g.replay(); stream.synchronize()
        ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
        ev0.record(stream)
        for _ in range(reps): g.replay()
        ev1.record(stream); stream.synchronize()
    return ev0.elapsed_time(ev1) / (steps * reps) * 1e3

if __name__ == "__main__":
  for label, fixture, B, E in (('C3 2023 schema', 'g2023_p2', None, 65536), ('2020 schema', 'g2020_cz1', None, 65536),
                               ('C4 synthetic (2020 devices)', 'g2020_cz1', 1024, 1024), ('C4 synthetic (2022 devices)', 'g2022_all', 1024, 1024)):
      spec = golden(fixture).spec()
      if B: spec = tile_district(spec, B)
      tab = spec.episode_tables(0)
      eng = StepEngine(tab, E)
      low, high = spec.action_limits()
      lo, hi = torch.from_numpy(low).cuda(), torch.from_numpy(high).cuda()
      acts = [lo[:, None] + torch.rand((len(low), E), device='cuda') * (hi - lo)[:, None] for _ in range(2)]
      us = measure(eng, acts)
      units = E * eng.n_bldg; bpu = eng.algorithmic_bytes_per_unit()
      print(f'{label}: B={eng.n_bldg} E={E} lean={eng.lean}: {us:.1f} us/step  {units/us*1e6:.3e} building-timesteps/s  {units*bpu/us/1e3:.0f} GB/s ({bpu:.1f} B/unit)', flush=True)
