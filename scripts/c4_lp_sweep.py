"""C4 shard (1024 buildings x 1024 envs, 2020 device set): chunk geometry with the LDS-staged parameter blocks (GPU box)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests')); sys.path.insert(0, str(ROOT / 'scripts'))
import torch
from golden_util import golden
from citylearn_amd.engine import StepEngine
from citylearn_amd.synthetic import tile_district
from c4_bench import measure
E = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
spec = tile_district(golden('g2020_cz1').spec(), 1024)
tab = spec.episode_tables(0)
low, high = spec.action_limits()
lo, hi = torch.from_numpy(low).cuda(), torch.from_numpy(high).cuda()
acts = [lo[:, None] + torch.rand((len(low), E), device='cuda') * (hi - lo)[:, None] for _ in range(2)]
for fv in (0, 3):
    for vec in (1, 2):
        for bc, nw in ((0, 0), (16, 16), (32, 16), (48, 16), (64, 16), (24, 12), (36, 12)):
            try:
                eng = StepEngine(tab, E, tuning=dict(vec=vec, nw=nw, b_chunk=bc, full_variant=fv))
                us = measure(eng, acts, steps=40, reps=4)
                print(f'1024 x {E} params in {"LDS" if fv == 0 else "SGPRs"} vec={vec} b_chunk={bc} nw={nw}: {us:.2f} us', flush=True)
            except Exception as e:
                print(f'vec={vec} b_chunk={bc} nw={nw}: {type(e).__name__} {str(e)[:80]}', flush=True)
