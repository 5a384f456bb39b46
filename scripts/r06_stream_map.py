"""The streaming regime beyond the round-6 rule: env-major (as selected, and with non-temporal stores) against the building-major kernel with non-temporal stores
for 6 .. 20 buildings x 524 288 .. 2 097 152 envs, both precision models -- engines alive side by side per cell, three round-robin rounds, medians.
Usage: r06_stream_map.py out.jsonl"""
import json
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

base = load_district(sample_schema('citylearn_challenge_2022_phase_all_720h'))
out = open(sys.argv[1], 'a')
VARIANTS = (('default', {}), ('env-major nt', dict(envmajor=1, nt_stores=1)), ('env-major plain', dict(envmajor=1, nt_stores=2)),
            ('lean4 nt', dict(envmajor=2, lean_variant=2, nt_stores=1)), ('lean4 plain', dict(envmajor=2, lean_variant=2, nt_stores=2)))
for B in (tuple(int(x) for x in sys.argv[2].split(",")) if len(sys.argv) > 2 else (6, 9, 12, 17, 20)):
    spec = tile_district(base, B, jitter=0.0 if B <= 17 else 0.1)
    tab = spec.episode_tables(0)
    low, high = spec.action_limits()
    lo, hi = torch.from_numpy(low).cuda(), torch.from_numpy(high).cuda()
    for E in (393216, 524288, 786432, 1048576, 1572864, 2097152):
        acts = [lo[:, None] + torch.rand((len(low), E), device='cuda') * (hi - lo)[:, None] for _ in range(2)]
        for prec, label in ((False, 'fp32'), ('chain', 'chain')):
            engs = {}
            for name, tun in VARIANTS:
                engs[name] = StepEngine(tab, E, f64_maps=prec, tuning=tun)
                engs[name].trace_kernels()
            res = {k: [] for k in engs}
            for rnd in range(3):
                for name, eng in engs.items():
                    res[name].append(measure(eng, acts, steps=20, reps=2))
            row = {'B': B, 'E': E, 'units_M': round(B * E / 2 ** 20, 2), 'precision': label, 'default_kernel': engs['default'].last_kernels,
                   **{k: round(sorted(v)[1], 2) for k, v in res.items()}}
            row['best'] = min((k for k in res if k != 'default'), key=lambda k: row[k])
            row['default_vs_best'] = round(row['default'] / row[row['best']], 3)
            print(json.dumps(row), flush=True)
            out.write(json.dumps(row) + '\n'); out.flush()
            del engs
            torch.cuda.empty_cache()
        del acts
        torch.cuda.empty_cache()
