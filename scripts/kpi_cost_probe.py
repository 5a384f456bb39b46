"""What the streaming KPI accumulators cost per step (GPU box)."""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests')); sys.path.insert(0, str(ROOT / 'scripts'))
from golden_util import golden
import os
from citylearn_amd import _lib
if os.environ.get('CL_ALT_LIB'):
    _lib.LIB_PATH = Path(os.environ['CL_ALT_LIB']).resolve()
from citylearn_amd.engine import StepEngine
from c4_bench import measure
for name, E in (('g2022_all', 65536), ('g2020_cz1', 65536)):
    spec = golden(name).spec(); tab = spec.episode_tables(0)
    low, high = spec.action_limits()
    lo, hi = torch.from_numpy(low).cuda(), torch.from_numpy(high).cuda()
    acts = [lo[:, None] + torch.rand((len(low), E), device='cuda') * (hi - lo)[:, None] for _ in range(2)]
    for kpi in (False, True):
        eng = StepEngine(tab, E, kpi=kpi, detail=(kpi and len(sys.argv) > 1))
        us = min(measure(eng, acts, steps=40, reps=4) for _ in range(2))
        print(f'{name} {eng.n_bldg} x {E} kpi={kpi} (detail planes {kpi and len(sys.argv) > 1}): {us:.2f} us per step', flush=True)
        del eng
