"""C4 shards (1024 buildings x 1024 envs) with the deferred finish (cl_tuning.finish = 3): launch geometry sweep -- envs per lane x buildings
per workgroup row -- next to the second launch per step (finish = 1).  The round-2 geometry rules were tuned for the two launches.  GPU box."""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests')); sys.path.insert(0, str(ROOT / 'scripts'))
from golden_util import golden
from citylearn_amd.engine import StepEngine
from citylearn_amd.synthetic import tile_district
from c4_bench import measure

for label, fixture, combos in (('battery + PV', 'g2022_all', [(4, 16), (4, 32), (2, 16), (2, 32), (2, 64), (1, 16), (1, 32), (1, 64)]),
                               ('thermal', 'g2020_cz1', [(2, 32), (2, 16), (2, 64), (1, 16), (1, 32), (1, 64)])):
    spec = tile_district(golden(fixture).spec(), 1024)
    tab = spec.episode_tables(0)
    E = 1024
    low, high = spec.action_limits()
    lo, hi = torch.from_numpy(low).cuda(), torch.from_numpy(high).cuda()
    acts = [lo[:, None] + torch.rand((len(low), E), device='cuda') * (hi - lo)[:, None] for _ in range(2)]
    rows = []
    for fin in (3, 1):
        for vec, bc in [(0, 0)] + combos:
            tun = dict(finish=fin)
            if vec:
                tun.update(vec=vec, b_chunk=bc, nw=16)
            try:
                eng = StepEngine(tab, E, tuning=tun); eng.trace_kernels()
                us = min(measure(eng, acts, steps=60, reps=4) for _ in range(2))
                deferred = 'cl_finish_kernel' not in eng.last_kernels
                rows.append(f'finish={fin} ' + ('default' if not vec else f'{vec}/lane x {bc} bldg/row') + f': {us:.2f}' + ('' if deferred or fin == 1 else ' (NOT deferred)'))
                del eng
            except Exception as e:
                rows.append(f'finish={fin} {vec}/lane x {bc}: {type(e).__name__}')
    print(f'{label}:\n   ' + '\n   '.join(rows), flush=True)
