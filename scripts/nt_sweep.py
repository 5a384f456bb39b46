"""Non-temporal plane stores on / off by launch size (GPU box): decides CL_NT_MAX_UNITS in csrc/cl_kernels.hip.
    python scripts/nt_sweep.py [lean] [thermal] [c4] [flex]"""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests')); sys.path.insert(0, str(ROOT / 'scripts'))
from golden_util import golden
from citylearn_amd.engine import StepEngine
from citylearn_amd.synthetic import tile_district
from c4_bench import measure

what = sys.argv[1:] or ['lean', 'thermal', 'c4']
SHAPES = []
if 'lean' in what:
    SHAPES += [('lean 2022', 'g2022_all', None, e) for e in (16384, 65536, 131072, 196608, 262144, 524288, 1048576)]
if 'big' in what:
    SHAPES += [('lean 2022', 'g2022_all', None, e) for e in (524288, 655360, 786432, 1048576, 1572864)] + [('2020 thermal', 'g2020_cz1', None, e) for e in (524288, 1048576)]
if 'thermal' in what:
    SHAPES += [('2020 thermal', 'g2020_cz1', None, e) for e in (65536, 131072, 262144, 524288)] + [('C3 2023', 'g2023_p2', None, e) for e in (65536, 262144, 1048576)]
if 'c4' in what:
    SHAPES += [('C4 thermal 1024 bldgs', 'g2020_cz1', 1024, 1024), ('C4 thermal 1024 bldgs', 'g2020_cz1', 1024, 4096), ('C4 lean 1024 bldgs', 'g2022_all', 1024, 1024)]
for label, fixture, B, E in SHAPES:
    spec = golden(fixture).spec()
    if B: spec = tile_district(spec, B)
    tab = spec.episode_tables(0)
    low, high = spec.action_limits()
    lo, hi = torch.from_numpy(low).cuda(), torch.from_numpy(high).cuda()
    acts = [lo[:, None] + torch.rand((len(low), E), device='cuda') * (hi - lo)[:, None] for _ in range(2)]
    res = {}
    for nt in (2, 1, 2, 1):
        eng = StepEngine(tab, E, tuning=dict(nt_stores=nt))
        us = measure(eng, acts, steps=40 if E * eng.n_bldg < (8 << 20) else 12, reps=4)
        res.setdefault(nt, []).append(us)
        units = E * eng.n_bldg
        del eng
    plain, nt = min(res[2]), min(res[1])
    print(f'{label} {units // E} x {E} ({units / 2**20:.2f} Mi units): plain {plain:.2f} us, nt {nt:.2f} us  ({(nt / plain - 1) * 100:+.1f} %)', flush=True)
    del acts
    torch.cuda.empty_cache()
