#!/bin/bash
set -u
python - <<'PY'
import sys, torch
sys.path.insert(0, 'tests')
from golden_util import golden
from citylearn_amd.engine import StepEngine
g = golden('g2022_evs'); tab = g.spec().episode_tables(0)
for tun in ({}, {'flex_fused': 2}):
    eng = StepEngine(tab, 65536, reward='MARL', tuning=tun); eng.trace_kernels()
    a = (torch.rand((eng.n_act_cols, 65536), device='cuda') * 2 - 1).contiguous()
    eng.step(a, 1); print(tun, eng.last_kernels, 'lean', eng.lean)
PY
