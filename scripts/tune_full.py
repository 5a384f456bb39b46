"""Sweep launch geometry of the full (thermal) kernel on the GPU box: python scripts/tune_full.py"""
import sys, ctypes
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests'))
from golden_util import golden
from citylearn_amd import _lib
from citylearn_amd.engine import StepEngine
from citylearn_amd.synthetic import tile_district
from c4_bench import measure   # noqa

if __name__ == '__main__':
    for label, fixture, B, E, nws in (('2020', 'g2020_cz1', None, 65536, (9, 5, 3)), ('2023', 'g2023_p2', None, 65536, (3,)),
                                      ('C4-2020dev', 'g2020_cz1', 1024, 1024, (16,)), ('C4-lean', 'g2022_all', 1024, 1024, (16,))):
        spec = golden(fixture).spec()
        if B: spec = tile_district(spec, B)
        tab = spec.episode_tables(0)
        eng = StepEngine(tab, E)
        low, high = spec.action_limits()
        lo, hi = torch.from_numpy(low).cuda(), torch.from_numpy(high).cuda()
        acts = [lo[:, None] + torch.rand((len(low), E), device='cuda') * (hi - lo)[:, None] for _ in range(2)]
        units = E * eng.n_bldg; bpu = eng.algorithmic_bytes_per_unit()
        for vec in (1, 2, 4):
            for nw in nws:
                eng.tuning.vec, eng.tuning.nw = vec, (nw if not B else 0)
                us = measure(eng, acts)
                print(f'{label} vec={vec} nw={nw}: {us:.1f} us/step {units*bpu/us/1e3:.0f} GB/s', flush=True)
