"""Observation epilogue timing (GPU box): cl_observe_f32 alone and the step + observe pair (SURVEY 8d mode A-obs)."""
import sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests'))
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
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


x = torch.empty((65536, 476), device='cuda'); y = torch.empty_like(x)
us = timed(lambda: x.fill_(1.0)); print(f'torch fill_ of {x.numel()*4/1e6:.0f} MB: {us:.1f} us  {x.numel()*4/us/1e3:.0f} GB/s written')
us = timed(lambda: y.copy_(x)); print(f'torch copy_ of {x.numel()*4/1e6:.0f} MB: {us:.1f} us  {2*x.numel()*4/us/1e3:.0f} GB/s read+written')
del x, y
x = torch.empty((262144, 476), device='cuda')
us = timed(lambda: x.fill_(1.0), n=20); print(f'torch fill_ of {x.numel()*4/1e6:.0f} MB: {us:.1f} us  {x.numel()*4/us/1e3:.0f} GB/s written')
del x

for name, E in (('g2022_all', 65536), ('g2022_all', 262144), ('g2020_cz1', 65536), ('g2023_p2', 65536)):
    g = golden(name); spec = g.spec(); tab = spec.episode_tables(0)
    for normalize in (False, True):
        lay = ObservationLayout(spec, 'current', normalize)
        ot = lay.episode(tab)
        eng = StepEngine(tab, E, detail=ot.needs_detail or any(b.is_dynamics for b in spec.buildings))
        stage = None
        if any(b.is_dynamics for b in spec.buildings):
            from citylearn_amd.dynamics import LSTMStage
            stage = LSTMStage(spec, tab, eng)
        w = ObservationWriter(eng, ot, stage)
        acts = torch.rand((eng.n_act_cols, E), device='cuda') * 2 - 1
        eng.tuning.obs_variant = 1; us_row = timed(lambda: w.write(7))
        eng.tuning.obs_variant = 5; us_row1 = timed(lambda: w.write(7))
        eng.tuning.obs_variant = 3; us_wave = timed(lambda: w.write(7))
        alt = []
        for rows in (8, 16, 64):
            eng.tuning.obs_variant, eng.tuning.obs_rows = 2, rows; alt.append(f'{rows}: {timed(lambda: w.write(7)):.1f}')
        eng.tuning.obs_variant, eng.tuning.obs_rows = 0, 0
        us = timed(lambda: w.write(7))
        us0 = timed(lambda: w.write(0))
        rowt = w.table[7]
        dense = torch.empty((E, w.n_cols), device='cuda')
        us_t = timed(lambda: dense.copy_(rowt.expand(E, -1)))
        print(f'   (all-exogenous write(0): {us0:.1f} us; torch broadcast copy_ of the row: {us_t:.1f} us)')
        del dense
        by = w.algorithmic_bytes()
        print(f'   (round-1 row-wise kernel: {us_row:.1f} us; one-round-trip row-wise kernel: {us_row1:.1f} us; wave-independent kernel: {us_wave:.1f} us; tile kernel by block rows: {", ".join(alt)} us)')
        dep_tables, dep_cols = ot.compact()
        wc = ObservationWriter(eng, dep_tables, stage)
        us_c = timed(lambda: wc.write(7))
        both_c = timed(lambda: (eng.step(acts, 7), wc.write(8)))
        print(f'   compact form (shared row + [E, {len(dep_cols)}] dependent matrix, {wc.algorithmic_bytes() / 1e6:.1f} MB): observe {us_c:.1f} us | '
              f'step+observe {both_c:.1f} us  {eng.n_bldg*E/both_c*1e6:.3e} building-timesteps/s')
        both = timed(lambda: (eng.step(acts, 7), w.write(8)))
        step_b = eng.algorithmic_bytes_per_unit() * eng.n_bldg * E
        print(f'{name} E={E} n_cols={w.n_cols} dep={ot.n_dependent} norm={normalize}: observe {us:.1f} us  {by/us/1e3:.0f} GB/s '
              f'({by/us/1e3/8000*100:.1f}% of 8 TB/s) | step+observe {both:.1f} us  {(by+step_b)/both/1e3:.0f} GB/s  '
              f'{eng.n_bldg*E/both*1e6:.3e} building-timesteps/s ({(by+step_b)/(eng.n_bldg*E):.0f} B/unit)', flush=True)
        del eng, w, stage
        torch.cuda.empty_cache()
