"""Chunk geometry of the building-chunked launches (b_chunk x waves per workgroup x envs per lane) on the C4 shards, and waves per
workgroup on the non-chunked thermal shapes (GPU box)."""
import sys, os
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests')); sys.path.insert(0, str(ROOT / 'scripts'))
import torch, ctypes
from golden_util import golden
from citylearn_amd.engine import StepEngine
from citylearn_amd.synthetic import tile_district
from c4_bench import measure
for label, fixture, B, E in (('2020 schema', 'g2020_cz1', None, 65536), ('C4 2020 devices', 'g2020_cz1', 1024, 1024), ('C4 2022 devices', 'g2022_all', 1024, 1024), ('C3 2023', 'g2023_p2', None, 65536)):
    spec = golden(fixture).spec()
    if B: spec = tile_district(spec, B)
    tab = spec.episode_tables(0)
    low, high = spec.action_limits()
    lo, hi = torch.from_numpy(low).cuda(), torch.from_numpy(high).cuda()
    acts = [lo[:, None] + torch.rand((len(low), E), device='cuda') * (hi - lo)[:, None] for _ in range(2)]
    for vec in (0, 1, 2, 4):
        # chunked launches (B > 32): buildings per workgroup row x waves per workgroup; others: waves per workgroup
        for bc, nw in (((0, 0), (8, 8), (16, 8), (16, 16), (32, 16), (32, 8), (64, 16)) if B else ((0, 0), (0, 3), (0, 5), (0, 9), (0, 16))):
            eng = StepEngine(tab, E, tuning=dict(vec=vec, nw=nw, b_chunk=bc))
            try:
                us = measure(eng, acts, steps=40, reps=4)
                print(f'{label} vec={vec} b_chunk={bc} nw={nw}: {us:.2f} us', flush=True)
            except Exception as e:
                print(f'{label} vec={vec} b_chunk={bc} nw={nw}: {type(e).__name__} {str(e)[:80]}', flush=True)
