"""Cost of CLD_F64_MAPS (battery map in the reference's mixed float64 / float32 precision) next to the default fp32 kernels: us per
env step, hipGraph replay, same box, alternating.  GPU box."""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from citylearn_amd import load_district
from citylearn_amd.data import sample_schema
from citylearn_amd.engine import StepEngine


def measure(eng, acts, steps=100, reps=5):
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        for t in range(3):
            eng.step(acts[t % 2], 1 + t)
        stream.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            for t in range(steps):
                eng.step(acts[t % 2], 1 + t % 600)
        g.replay(); stream.synchronize()
        ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
        ev0.record(stream)
        for _ in range(reps):
            g.replay()
        ev1.record(stream); stream.synchronize()
    return ev0.elapsed_time(ev1) / (steps * reps) * 1e3


def main():
    for label, name, E in (('2022 (17 buildings, battery + PV)', 'citylearn_challenge_2022_phase_all_720h', 65536),
                           ('2020 cz1 (9 buildings, thermal)', 'citylearn_challenge_2020_climate_zone_1_744h', 65536),
                           ('2023 p2 (3 buildings, outage)', 'citylearn_challenge_2023_phase_2_local_evaluation_720h', 65536),
                           ('2022 x 262144', 'citylearn_challenge_2022_phase_all_720h', 262144)):
        spec = load_district(sample_schema(name))
        tab = spec.episode_tables(0)
        low, high = spec.action_limits()
        lo, hi = torch.from_numpy(low).cuda(), torch.from_numpy(high).cuda()
        acts = [lo[:, None] + torch.rand((len(low), E), device='cuda') * (hi - lo)[:, None] for _ in range(2)]
        res = {}
        for rnd in range(2):
            for f64 in (False, True):
                eng = StepEngine(tab, E, f64_maps=f64)
                eng.trace_kernels()
                us = measure(eng, acts)
                res.setdefault(f64, []).append(us)
                k = eng.last_kernels
                bpu = eng.algorithmic_bytes_per_unit()
                del eng
            torch.cuda.empty_cache()
        print(f'{label} x {E}: fp32 {res[False]} us   f64 maps {res[True]} us  ({k}, {bpu:.1f} B/unit)', flush=True)


if __name__ == '__main__':
    main()
