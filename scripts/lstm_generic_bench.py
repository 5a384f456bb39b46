"""Timing line of the fallback LSTM kernel (`cl_lstm_generic_kernel`: shapes the matrix-core kernel does not cover) -- baeda_3dem, whose
Building_4 is a one-layer LSTM(11 -> 50), and g2023_both, whose Building_1 takes both demands (LSTM(14 -> 16, 2 layers) on the generic path).
GPU box; stage only (cl_lstm_step_f32 + cl_lstm_generic_step_f32), eager, us per env step."""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests'))
from golden_util import golden
from citylearn_amd.engine import StepEngine
from citylearn_amd.dynamics import LSTMStage

for name in ('s_baeda', 'g2023_both', 'g2023_p2'):
    g = golden(name); spec = g.spec(); tab = spec.episode_tables(0)
    for E in (4096, 65536):
        eng = StepEngine(tab, E, detail=True)
        stage = LSTMStage(spec, tab, eng)
        B = eng.n_bldg
        cd = torch.rand((B, E), device='cuda') * 3
        for t in range(12, 16):
            stage.step(t, cd, cd)
        torch.cuda.synchronize()
        ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
        n = 10
        ev0.record()
        for t in range(20, 20 + n):
            stage.step(t, cd, cd)
        ev1.record(); torch.cuda.synchronize()
        us = ev0.elapsed_time(ev1) / n * 1e3
        gen = 'none' if stage.generic is None else f'H = {stage.generic["h"]}'
        print(f'{name}: {B} buildings x {E} envs, generic-kernel buildings: {gen}: {us:.1f} us per LSTM stage step ({B * E / us * 1e6:.3e} building-timesteps/s)', flush=True)
        del eng, stage
        torch.cuda.empty_cache()
