"""What the streaming KPI accumulators cost per step (GPU box): no KPIs / in the step launch (default, and with the step-only wave counts) /
as a launch after the step (`kpi_passes = 1`), several shapes.  Output tracked as profiles/r03_kpi_in_step_probe.log."""
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
for name, E in (('g2022_all', 65536), ('g2020_cz1', 65536), ('g2020_cz1', 16384), ('g2020_cz1', 262144), ('g2023_p2', 65536)):
    spec = golden(name).spec(); tab = spec.episode_tables(0)
    low, high = spec.action_limits()
    lo, hi = torch.from_numpy(low).cuda(), torch.from_numpy(high).cuda()
    acts = [lo[:, None] + torch.rand((len(low), E), device='cuda') * (hi - lo)[:, None] for _ in range(2)]
    for kpi, tuning in ((False, None), (True, None), (True, dict(nw=5)), (True, dict(nw=9)), (True, dict(kpi_passes=1)), (True, None), (False, None)):
        if tuning and name == 'g2022_all':
            continue
        eng = StepEngine(tab, E, kpi=kpi, detail=(kpi and len(sys.argv) > 1), tuning=tuning)
        eng.trace_kernels()
        us = min(measure(eng, acts, steps=40, reps=4) for _ in range(2))
        print(f'{name} {eng.n_bldg} x {E} kpi={kpi} (detail planes {eng.detail}) {eng.last_kernels}: {us:.2f} us per step', flush=True)
        del eng
