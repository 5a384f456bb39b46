"""Mode B: envs per lane by batch size (GPU box)."""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests'))
from golden_util import golden
from citylearn_amd.engine import StepEngine
g = golden('g2022_all'); spec = g.spec(); tab = spec.episode_tables(0)
low, high = spec.action_limits()
K = 24
for E in (32768, 49152, 65536, 98304, 131072):
    res = []
    for vec in (1, 2):
        eng = StepEngine(tab, E, tuning=dict(vec=vec)); eng.set_action_limits(low, high)
        ret = torch.zeros(E, device='cuda')
        for i in range(3): eng.rollout(K, seed=i, ret_env=ret, t0=1)
        torch.cuda.synchronize()
        ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
        n = 20
        ev0.record()
        for i in range(n): eng.rollout(K, seed=i, ret_env=ret, t0=1 + (i * K) % 600)
        ev1.record(); torch.cuda.synchronize()
        res.append(f'vec={vec}: {ev0.elapsed_time(ev1) / n / K * 1e3:.2f} us/step')
        del eng
    print(f'17 x {E}, K={K}: ' + ' | '.join(res), flush=True)
