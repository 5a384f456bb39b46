"""cl_step_full_tp_kernel with more tiles per workgroup at larger batches (GPU box)."""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests')); sys.path.insert(0, str(ROOT / 'scripts'))
from golden_util import golden
from citylearn_amd.engine import StepEngine
from c4_bench import measure
from tp_sweep import district
for B in (9, 6, 3):
    spec = district(B); tab = spec.episode_tables(0)
    low, high = spec.action_limits()
    lo, hi = torch.from_numpy(low).cuda(), torch.from_numpy(high).cuda()
    for E in (65536, 131072, 262144):
        acts = [lo[:, None] + torch.rand((len(low), E), device='cuda') * (hi - lo)[:, None] for _ in range(2)]
        res = []
        for label, tun in [('default', dict())] + [(f'{tp} tiles x {v}/lane', dict(full_variant=5, vec=v, b_chunk=tp)) for v, tp in ((2, 2), (2, 4), (2, 8), (1, 4), (1, 8))]:
            try:
                eng = StepEngine(tab, E, tuning=tun)
                us = min(measure(eng, acts, steps=40, reps=4) for _ in range(2))
                res.append(f'{label}: {us:.2f}')
                del eng
            except Exception as e:
                res.append(f'{label}: {type(e).__name__}')
        print(f'B={B} E={E}: ' + ' | '.join(res) + ' us', flush=True)
        del acts
        torch.cuda.empty_cache()
