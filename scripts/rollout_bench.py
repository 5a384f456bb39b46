"""Mode-B timing on the GPU box: fused K-step rollouts with the on-device Philox policy (BASELINE config 5 shape per GPU:
17 buildings x 32768 envs, K = 24) and at the headline batch."""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests'))
from golden_util import golden
import os
from citylearn_amd import _lib
if os.environ.get('CL_ALT_LIB'):           # A/B experiments: a second build of the library
    _lib.LIB_PATH = Path(os.environ['CL_ALT_LIB']).resolve()
from citylearn_amd.engine import StepEngine

g = golden('g2022_all'); spec = g.spec(); tab = spec.episode_tables(0)
low, high = spec.action_limits()
for E in [int(x) for x in sys.argv[1:]] or (32768, 65536, 262144):
    for K in (24, 96):
        eng = StepEngine(tab, E); eng.set_action_limits(low, high)
        ret = torch.zeros(E, device='cuda')
        for i in range(3): eng.rollout(K, seed=i, ret_env=ret, t0=1)
        torch.cuda.synchronize()
        ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
        n = 20
        ev0.record()
        for i in range(n): eng.rollout(K, seed=i, ret_env=ret, t0=1 + (i * K) % 600)
        ev1.record(); torch.cuda.synchronize()
        ms = ev0.elapsed_time(ev1) / n
        print(f'E={E} K={K}: {ms*1e3:.1f} us/launch  {ms*1e3/K:.2f} us/step  {E*17*K/ms*1e3:.3e} building-timesteps/s', flush=True)
