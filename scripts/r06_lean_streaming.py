"""The building-major lean kernel forced onto the HBM-streaming shape (17 x 1 048 576, cl_tuning.lean_variant = 2) by store hint and envs per lane, next to
the env-major kernel the library selects there -- with CITYLEARN_AMD_LIB pointing at a -DCL_EXP_NTL_ALL build the same with the non-temporal hint on every load."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'scripts'))
from citylearn_amd import load_district
from citylearn_amd.data import sample_schema
from citylearn_amd.engine import StepEngine
from f64_cost import measure

E = 1048576
spec = load_district(sample_schema('citylearn_challenge_2022_phase_all_720h'))
tab = spec.episode_tables(0)
low, high = spec.action_limits()
lo, hi = torch.from_numpy(low).cuda(), torch.from_numpy(high).cuda()
acts = [lo[:, None] + torch.rand((len(low), E), device='cuda') * (hi - lo)[:, None] for _ in range(2)]
for prec, label in ((False, 'fp32'), ('chain', 'chain')):
    for tun in ({}, dict(envmajor=2, lean_variant=2, nt_stores=1), dict(envmajor=2, lean_variant=2, nt_stores=2), dict(envmajor=2, lean_variant=2, nt_stores=1, vec=2), dict(envmajor=2, lean_variant=2, nt_stores=2, vec=2),
                dict(envmajor=2, lean_variant=2, nt_stores=2, nw=16), dict(envmajor=2, lean_variant=2, nt_stores=1, nw=16)):
        try:
            eng = StepEngine(tab, E, f64_maps=prec, tuning=tun)
            eng.trace_kernels()
            us = sorted(measure(eng, acts, steps=20, reps=3) for _ in range(3))[1]
            print(f'{label:5s} {str(tun):60s} {us:7.2f} us  {eng.last_kernels}', flush=True)
            del eng
        except Exception as e:
            print(f'{label:5s} {str(tun):60s} {str(e)[:90]}', flush=True)
