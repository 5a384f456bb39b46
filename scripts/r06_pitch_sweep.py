"""Row pitch of the state / output planes on the HBM-streaming shape (17 x 1 048 576): us per step for pads of 0 .. 8 448 envs beyond n_env, engines alive side
by side in ONE process and measured round-robin (a process lands somewhere in a +- 4 % band: LAB_NOTES 5.3), default precision model and the fp32 map.
Usage: r06_pitch_sweep.py [chain | fp32]"""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'scripts'))
from citylearn_amd import load_district
from citylearn_amd.data import sample_schema
from citylearn_amd.engine import StepEngine
from f64_cost import measure

prec = {'chain': 'chain', 'fp32': False}[sys.argv[1] if len(sys.argv) > 1 else 'chain']
E = 1048576
spec = load_district(sample_schema('citylearn_challenge_2022_phase_all_720h'))
tab = spec.episode_tables(0)
low, high = spec.action_limits()
lo, hi = torch.from_numpy(low).cuda(), torch.from_numpy(high).cuda()
acts = [lo[:, None] + torch.rand((len(low), E), device='cuda') * (hi - lo)[:, None] for _ in range(2)]
pads = tuple(int(x) for x in sys.argv[2].split(",")) if len(sys.argv) > 2 else (0, 64, 128, 256, 320, 512, 768, 1024, 1280, 2048 + 256, 4096 + 256, 8192 + 256)
res = {p: [] for p in pads}
for rnd in range(int(sys.argv[3]) if len(sys.argv) > 3 else 3):
    for p in pads:
        eng = StepEngine(tab, E, f64_maps=prec, env_pitch=E + p)
        res[p].append(measure(eng, acts, steps=20, reps=3))
        del eng
        torch.cuda.empty_cache()
for p in pads:
    print(f'pad {p:5d} envs: ' + ' '.join(f'{u:7.2f}' for u in res[p]) + f'  us per step  (median {sorted(res[p])[len(res[p]) // 2]:.2f})', flush=True)
