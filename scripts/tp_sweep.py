"""Thermal kernel with several env tiles per workgroup (cl_step_full_tp_kernel) against the one-tile kernel (GPU box): comparison of the
results, then a size sweep.   python scripts/tp_sweep.py [compare] [sweep]"""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests')); sys.path.insert(0, str(ROOT / 'scripts'))
from golden_util import golden
from citylearn_amd import abi
from citylearn_amd.engine import StepEngine
from citylearn_amd.synthetic import tile_district
from c4_bench import measure
what = (sys.argv[1:] or ["compare", "sweep"]) if __name__ == "__main__" else []


def district(B):
    if B == 6: return golden('s_2023_p3').spec()
    if B == 3: return golden('g2023_p2').spec()
    spec = golden('g2020_cz1').spec()
    return spec if B == 9 else tile_district(spec, B)


if 'compare' in what:
    for B, E in ((9, 4996), (6, 516), (3, 260), (16, 1028)):
        spec = district(B); tab = spec.episode_tables(0)
        low, high = spec.action_limits()
        lo, hi = torch.from_numpy(low).cuda()[:, None], torch.from_numpy(high).cuda()[:, None]
        for kind in ('RewardFunction', 'MARL', 'SolarPenaltyReward', 'IndependentSACReward'):
            ref = StepEngine(tab, E, reward=kind, tuning=dict(full_variant=3, vec=1))
            for vec in (1, 2):
                tp = StepEngine(tab, E, reward=kind, tuning=dict(full_variant=5, vec=vec))
                gen = torch.Generator(device='cuda').manual_seed(B)
                ref.reset(); worst = 0.0; same = True
                for t in range(60):
                    a = (lo + torch.rand((ref.n_act_cols, E), device='cuda', generator=gen) * (hi - lo)).contiguous()
                    ref.step(a, t); tp.step(a, t)
                    same &= torch.equal(ref.state, tp.state) and torch.equal(ref.out_bldg[abi.CLO_NET], tp.out_bldg[abi.CLO_NET])
                    if kind != 'MARL': same &= torch.equal(ref.out_bldg[abi.CLO_REWARD], tp.out_bldg[abi.CLO_REWARD])
                    d = ((ref.out_env - tp.out_env).abs() / (1e-4 + 1e-4 * ref.out_env.abs())).max().item()
                    dr = ((ref.out_bldg[abi.CLO_REWARD] - tp.out_bldg[abi.CLO_REWARD]).abs() / (1e-4 + 1e-4 * ref.out_bldg[abi.CLO_REWARD].abs())).max().item()
                    worst = max(worst, d, dr)
                    tp.state.copy_(ref.state)
                print(f'B={B} E={E} {kind} vec={vec}: building planes identical: {same}; district sums / MARL rewards worst {worst:.2e} of the 1e-4 tolerance', flush=True)

if 'sweep' in what:
    for B in (9, 6, 12, 16):
        spec = district(B); tab = spec.episode_tables(0)
        low, high = spec.action_limits()
        lo, hi = torch.from_numpy(low).cuda(), torch.from_numpy(high).cuda()
        for E in (16384, 32768, 65536, 131072, 262144):
            acts = [lo[:, None] + torch.rand((len(low), E), device='cuda') * (hi - lo)[:, None] for _ in range(2)]
            res = []
            for label, tun in (('one tile', dict()), ('2 tiles x 2 envs/lane, 16 waves', dict(full_variant=5, vec=2)),
                               (f'2 tiles x 2 envs/lane, {min(16, 2 * B)} waves', dict(full_variant=5, vec=2, nw=min(16, 2 * B))),
                               ('4 tiles x 1 env/lane, 16 waves', dict(full_variant=5, vec=1))):
                eng = StepEngine(tab, E, tuning=tun)
                us = min(measure(eng, acts, steps=40, reps=4) for _ in range(2))
                res.append(f'{label}: {us:.2f}')
                del eng
            print(f'B={B} E={E}: ' + ' | '.join(res) + ' us', flush=True)
            del acts
            torch.cuda.empty_cache()
