"""Kernel-variant sweep on the GPU box (hipGraph replay so the host launch rate does not mask the kernel):
python scripts/tune.py [E ...]"""
import sys, ctypes
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests'))
from golden_util import golden
from citylearn_amd import _lib
from citylearn_amd.engine import StepEngine

def measure(eng, acts, steps=100, reps=8):
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        for t in range(5): eng.step(acts[t % 4], 1 + t)
        stream.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            for t in range(steps): eng.step(acts[t % 4], 1 + t % 700)
        g.replay(); stream.synchronize()
        ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
        ev0.record(stream)
        for _ in range(reps): g.replay()
        ev1.record(stream); stream.synchronize()
    return ev0.elapsed_time(ev1) / (steps * reps) * 1e3   # us per step

if __name__ == '__main__':
    g = golden('g2022_all'); spec = g.spec(); tab = spec.episode_tables(0)
    Es = [int(x) for x in sys.argv[1:]] or [65536]
    for E in Es:
        eng = StepEngine(tab, E)
        acts = [(torch.rand((eng.n_act_cols, E), device='cuda') * 2 - 1) for _ in range(4)]
        bpu = eng.algorithmic_bytes_per_unit(); units = E * eng.n_bldg
        res = []
        for pipe in (0, 1):
            for vec in (1, 2, 4):
                for nw in (16, 9, 6):
                    eng.tuning.vec, eng.tuning.no_chunks, eng.tuning.nw = vec, pipe, nw
                    us = measure(eng, acts)
                    res.append((us, pipe, vec, nw))
                    print(f'E={E} pipe={pipe} vec={vec} nw={nw}: {us:.2f} us/step  {units*bpu/us/1e3:.0f} GB/s', flush=True)
        print('best', min(res))
