"""Env-major against building-major (latency-ordered lean) kernel on 17 buildings from 131 072 to 2 097 152 envs, both precision models, one process,
alternating: where does the rule `env-major above 122 880 envs` stop paying?  Usage: r06_lean_vs_envmajor.py [n_bldg [E,E,...]]"""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'scripts'))
from citylearn_amd import load_district
from citylearn_amd.data import sample_schema
from citylearn_amd.engine import StepEngine
from citylearn_amd.synthetic import tile_district
from f64_cost import measure

B = int(sys.argv[1]) if len(sys.argv) > 1 else 17
spec = load_district(sample_schema('citylearn_challenge_2022_phase_all_720h'))
if B != 17:
    spec = tile_district(spec, B, jitter=0.0 if B <= 17 else 0.1)
tab = spec.episode_tables(0)
low, high = spec.action_limits()
lo, hi = torch.from_numpy(low).cuda(), torch.from_numpy(high).cuda()
ES = tuple(int(x) for x in sys.argv[2].split(',')) if len(sys.argv) > 2 else (131072, 196608, 262144, 393216, 524288, 786432, 1048576, 2097152)
for E in ES:
    acts = [lo[:, None] + torch.rand((len(low), E), device='cuda') * (hi - lo)[:, None] for _ in range(2)]
    for prec, label in ((False, 'fp32'), ('chain', 'chain')):
        row = {}
        for name, tun in (('default', {}), ('env-major', dict(envmajor=1)), ('lean4', dict(envmajor=2, lean_variant=2)), ('lean4 nt', dict(envmajor=2, lean_variant=2, nt_stores=1))):
            eng = StepEngine(tab, E, f64_maps=prec, tuning=tun)
            eng.trace_kernels()
            us = sorted(measure(eng, acts, steps=20 if E >= 524288 else 50, reps=3) for _ in range(3))[1]
            row[name] = (us, eng.last_kernels)
            del eng
            torch.cuda.empty_cache()
        print(f'{B} x {E:8d} {label:5s} ' + '  '.join(f'{k}: {v[0]:7.2f}' for k, v in row.items()) + f'   [{row["default"][1]} | {row["lean4"][1]}]', flush=True)
    del acts
    torch.cuda.empty_cache()
