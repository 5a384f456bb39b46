"""cl_step_full_tp_kernel with ONE tile per workgroup at the batch sizes where that is one workgroup per CU (GPU box)."""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests')); sys.path.insert(0, str(ROOT / 'scripts'))
from citylearn_amd.engine import StepEngine
from c4_bench import measure
from tp_sweep import district
for B in (9, 6, 12, 16):
    spec = district(B); tab = spec.episode_tables(0)
    low, high = spec.action_limits()
    lo, hi = torch.from_numpy(low).cuda(), torch.from_numpy(high).cuda()
    for E, cfgs in ((16384, (('1 tile x 1/lane, B waves', dict(full_variant=5, vec=1, b_chunk=1, nw=min(16, B))), ('2 tiles x 1/lane', dict(full_variant=5, vec=1, b_chunk=2)))),
                    (32768, (('1 tile x 2/lane, B waves', dict(full_variant=5, vec=2, b_chunk=1, nw=min(16, B))), ('2 tiles x 1/lane', dict(full_variant=5, vec=1, b_chunk=2)),
                             ('4 tiles x 1/lane', dict(full_variant=5, vec=1, b_chunk=4))))):
        acts = [lo[:, None] + torch.rand((len(low), E), device='cuda') * (hi - lo)[:, None] for _ in range(2)]
        res = []
        for label, tun in (('default', dict()),) + cfgs:
            eng = StepEngine(tab, E, tuning=tun)
            res.append(f'{label}: {min(measure(eng, acts, steps=40, reps=4) for _ in range(2)):.2f}')
            del eng
        print(f'B={B} E={E}: ' + ' | '.join(res) + ' us', flush=True)
