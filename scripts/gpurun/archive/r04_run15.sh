#!/bin/bash
python - <<'PY' 2>&1 | grep -v amdgpu.ids
import sys, torch
sys.path.insert(0, 'tests'); sys.path.insert(0, 'scripts')
from citylearn_amd.engine import StepEngine
from c4_bench import measure
from tp_sweep import district
for B in (9, 6, 12):
    spec = district(B); tab = spec.episode_tables(0)
    low, high = spec.action_limits()
    lo, hi = torch.from_numpy(low).cuda(), torch.from_numpy(high).cuda()
    E = 65536
    acts = [lo[:, None] + torch.rand((len(low), E), device='cuda') * (hi - lo)[:, None] for _ in range(2)]
    res = []
    for label, tun in [('default', {}), ('one tile', dict(full_variant=3)), ('one tile nw=B', dict(full_variant=3, nw=B)), ('one tile nw=B vec=2', dict(full_variant=3, nw=B, vec=2)),
                       ('one tile vec=2', dict(full_variant=3, vec=2)), ('one tile nw=ceil(B/2) vec=2', dict(full_variant=3, nw=(B + 1) // 2, vec=2))]:
        try:
            eng = StepEngine(tab, E, tuning=tun); eng.trace_kernels()
            us = min(measure(eng, acts, steps=40, reps=4) for _ in range(3))
            res.append(f'{label}: {us:.2f} ({eng.last_kernels})')
            del eng
        except Exception as e:
            res.append(f'{label}: {type(e).__name__} {e}')
    print(f'B={B}:\n   ' + '\n   '.join(res), flush=True)
PY
