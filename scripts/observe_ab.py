"""Observation epilogue A/B (GPU box): CL_ALT_LIB=<lib> python scripts/observe_ab.py"""
import os, sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests')); sys.path.insert(0, str(ROOT / 'scripts'))
from citylearn_amd import _lib
if os.environ.get('CL_ALT_LIB'):
    _lib.LIB_PATH = Path(os.environ['CL_ALT_LIB']).resolve()
from golden_util import golden
from citylearn_amd.engine import StepEngine
from citylearn_amd.observations import ObservationLayout
from citylearn_amd.observe import ObservationWriter


def timed(fn, n=50, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (2 * n) * 1e3


for name, E in (('g2022_all', 65536), ('g2020_cz1', 65536), ('g2022_all', 262144)):
    spec = golden(name).spec(); tab = spec.episode_tables(0)
    lay = ObservationLayout(spec, 'current', False)
    ot = lay.episode(tab)
    eng = StepEngine(tab, E, detail=ot.needs_detail)
    w = ObservationWriter(eng, ot, None)
    acts = torch.rand((eng.n_act_cols, E), device='cuda') * 2 - 1
    us = sorted(timed(lambda: w.write(7)) for _ in range(3))[1]
    both = sorted(timed(lambda: (eng.step(acts, 7), w.write(8))) for _ in range(3))[1]
    dep_tables, dep_cols = ot.compact()
    wc = ObservationWriter(eng, dep_tables, None)
    both_c = sorted(timed(lambda: (eng.step(acts, 7), wc.write(8))) for _ in range(3))[1]
    print(f'{_lib.LIB_PATH.name} {name} E={E}: observe {us:.2f} us | step+observe {both:.2f} us | compact step+observe {both_c:.2f} us', flush=True)
    del eng, w, wc
    torch.cuda.empty_cache()
