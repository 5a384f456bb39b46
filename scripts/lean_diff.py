"""Which outputs differ between the lean kernel and the general kernel on the lean district (GPU box, debugging aid)."""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests'))
from golden_util import golden
from citylearn_amd import abi
from citylearn_amd.engine import StepEngine
tab = golden('g2022_all').spec().episode_tables(0)
E = 516
for kind in ('RewardFunction', 'MARL', 'SolarPenaltyReward', 'IndependentSACReward'):
    for vec in (1, 2, 4):
        e0, e1 = (StepEngine(tab, E, reward=kind, tuning=dict(vec=vec, lean_variant=v)) for v in (2, 1))
        gen = torch.Generator(device='cuda').manual_seed(vec)
        bad = {}
        for t in range(40):
            a = torch.rand((e0.n_act_cols, E), device='cuda', generator=gen) * 2 - 1
            e0.step(a, t); e1.step(a, t)
            for name, x, y in (('state', e0.state, e1.state), ('net', e0.out_bldg[0], e1.out_bldg[0]), ('reward', e0.out_bldg[1], e1.out_bldg[1]),
                               ('env_net', e0.out_env[0], e1.out_env[0]), ('env_cost', e0.out_env[1], e1.out_env[1]), ('env_em', e0.out_env[2], e1.out_env[2]),
                               ('env_rw', e0.out_env[3], e1.out_env[3])):
                if not torch.equal(x, y):
                    d = (x - y).abs()
                    bad.setdefault(name, [t, 0, 0.0]); bad[name][1] += int((d > 0).sum()); bad[name][2] = max(bad[name][2], float((d / (1e-30 + y.abs())).max()))
            e1.state.copy_(e0.state)
        print(kind, vec, bad if bad else 'identical', flush=True)
