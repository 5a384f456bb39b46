"""Step time of the EV district (2022 + EVs schema: 17 buildings, 8 chargers, 8 EVs, 1 washing machine) against the same
district without its flexible loads: what the extra `cl_flex_kernel` launch and the FLEX step instantiation cost.
Run on the GPU box: python scripts/ev_step_bench.py [n_env]"""
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / 'tests'))

from citylearn_amd import _lib                  # noqa: E402
import os                                       # noqa: E402
if os.environ.get('CL_ALT_LIB'):                # A/B experiments: a second build of the library
    _lib.LIB_PATH = Path(os.environ['CL_ALT_LIB']).resolve()
from citylearn_amd.engine import StepEngine     # noqa: E402
from golden_util import golden                  # noqa: E402

TUN = dict((k, int(v)) for k, v in (kv.split('=') for kv in os.environ.get('CL_TUNING', '').split(',') if kv))   # e.g. CL_TUNING=nt_stores=2


def timed(eng, a, steps=200, warm=20):
    T = eng.n_steps
    for i in range(warm):
        eng.step(a, (i % (T - 1)))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        eng.step(a, 1 + (i % (T - 2)))
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e6


def timed_graph(eng, a, steps=50, reps=20):
    """The same loop replayed as a hipGraph of `steps` consecutive env steps (what bench.py does for the headline): GPU time
    without the Python / launch cost of eager calls."""
    T = eng.n_steps
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        for i in range(5):
            eng.step(a, 1 + i)
        stream.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=stream):
            for i in range(steps):
                eng.step(a, 1 + (i % (T - 2)))
        gr.replay(); stream.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record(stream)
        for _ in range(reps):
            gr.replay()
        ev1.record(stream)
        stream.synchronize()
    return ev0.elapsed_time(ev1) / (reps * steps) * 1e3


def main():
    E = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    g = golden('g2022_evs')
    spec = g.spec()
    tab = spec.episode_tables(0)
    out = {'n_env': E}
    for reward in ('MARL', 'Electric_Vehicles_Reward_Function'):
        eng = StepEngine(tab, E, reward=reward, tuning=TUN)
        a = (torch.rand((eng.n_act_cols, E), device='cuda') * 2 - 1).contiguous()
        for fv in (1, 2, 4):
            eng.tuning.flex_vec = fv
            out[f'flex{fv}/{reward}'] = round(timed(eng, a), 2)
        eng.tuning.flex_vec = 0
        out[f'graph/{reward}'] = round(timed_graph(eng, a), 2)
    # K-step rollout with the on-device policy (cl_rollout_seq_f32: policy plane + flex + step + return per step), one graph
    eng = StepEngine(tab, E, reward='MARL', tuning=TUN)
    low, high = spec.action_limits()
    eng.set_action_limits(low, high)
    ret = torch.zeros(E, device='cuda')
    K, reps = 24, 20
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        eng.rollout(2, seed=1); eng.reset(); stream.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=stream):
            eng.rollout(K, seed=2, ret_env=ret, t0=1)
        gr.replay(); stream.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record(stream)
        for _ in range(reps):
            gr.replay()
        ev1.record(stream); stream.synchronize()
    out['graph/rollout24/MARL'] = round(ev0.elapsed_time(ev1) / (reps * K) * 1e3, 2)
    import copy
    plain = copy.copy(tab)
    plain.flex = None
    eng = StepEngine(plain, E, reward='MARL', n_act_cols=tab.flex.n_act_cols, tuning=TUN)
    a = (torch.rand((eng.n_act_cols, E), device='cuda') * 2 - 1).contiguous()
    out['no_flex/MARL'] = round(timed(eng, a), 2)
    out['graph/no_flex/MARL'] = round(timed_graph(eng, a), 2)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
