"""Env-major lean kernel (one wave = 64 envs x all buildings) against the building-major kernels: results and kernel time
(run under `rocprofv3 --kernel-trace --stats` for the durations).  GPU box."""
import sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests'))
from golden_util import golden
from citylearn_amd.engine import StepEngine

g = golden('g2022_all'); spec = g.spec(); tab = spec.episode_tables(0)
for reward in ('RewardFunction', 'MARL'):
    for E in (65536, 262144, 4096):
        e0 = StepEngine(tab, E, reward=reward); e1 = StepEngine(tab, E, reward=reward)
        lib = e0.lib
        acts = [(torch.rand((e0.n_act_cols, E), device='cuda') * 2 - 1).contiguous() for _ in range(4)]
        worst = 0.0
        for t in range(40):
            lib.cl_debug_set_envmajor(0); e0.step(acts[t % 4], t)
            lib.cl_debug_set_envmajor(1); e1.step(acts[t % 4], t)
        lib.cl_debug_set_envmajor(0)
        torch.cuda.synchronize()
        same = torch.equal(e0.state, e1.state) and torch.equal(e0.out_bldg[:2], e1.out_bldg[:2])
        d = (e0.out_env - e1.out_env).abs().max().item()
        print(f'{reward} E={E}: per-building planes identical: {same}; district sums max |diff| = {d:.3e} (different summation order)')
