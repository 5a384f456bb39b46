"""cl_step_full_tp_kernel with the software-pipelined item loop (round 4): waves per workgroup x tiles x envs per lane at the thermal
shapes (GPU box).  `default` = what the library picks."""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests')); sys.path.insert(0, str(ROOT / 'scripts'))
from citylearn_amd.engine import StepEngine
from c4_bench import measure
from tp_sweep import district
CASES = [('default', dict())] + [(f'{tp} tiles x {v}/lane x {nw} waves', dict(full_variant=5, vec=v, b_chunk=tp, nw=nw))
                                  for v, tp, nw in ((2, 2, 16), (2, 2, 12), (2, 2, 9), (2, 2, 6), (1, 4, 16), (1, 4, 12), (1, 4, 9), (1, 4, 8), (1, 4, 6))] \
        + [('one tile per workgroup', dict(full_variant=3))]
for B, sizes in ((9, (65536, 131072)), (6, (65536,)), (3, (65536,)), (12, (65536,)), (16, (65536,))):
    spec = district(B); tab = spec.episode_tables(0)
    low, high = spec.action_limits()
    lo, hi = torch.from_numpy(low).cuda(), torch.from_numpy(high).cuda()
    for E in sizes:
        acts = [lo[:, None] + torch.rand((len(low), E), device='cuda') * (hi - lo)[:, None] for _ in range(2)]
        res = []
        for label, tun in CASES:
            try:
                eng = StepEngine(tab, E, tuning=tun)
                us = min(measure(eng, acts, steps=40, reps=4) for _ in range(3))
                res.append(f'{label}: {us:.2f}')
                del eng
            except Exception as e:
                res.append(f'{label}: {type(e).__name__}')
        print(f'B={B} E={E}: ' + ' | '.join(res) + ' us', flush=True)
        del acts
        torch.cuda.empty_cache()
