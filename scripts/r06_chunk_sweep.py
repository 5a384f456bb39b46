"""Buildings per chunk of the building-chunked thermal step kernels away from the 1024-building district the rules were measured on:
us per step for B buildings x E envs with chunks of 32 / 64 / 128 / 256 (cl_tuning.b_chunk, 16 waves) next to the default.
Usage: r06_chunk_sweep.py out.jsonl [precision: chain | fp32]"""
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'scripts'))
from citylearn_amd import load_district
from citylearn_amd.data import sample_schema
from citylearn_amd.engine import StepEngine
from citylearn_amd.synthetic import tile_district
from f64_cost import measure


def main():
    out_path = sys.argv[1]
    prec = {'chain': 'chain', 'fp32': False}[sys.argv[2] if len(sys.argv) > 2 else 'chain']
    base = load_district(sample_schema('citylearn_challenge_2020_climate_zone_1_744h'))
    with open(out_path, 'a') as f:
        for B in (128, 256, 512, 1024):
            spec = tile_district(base, B)
            tab = spec.episode_tables(0)
            low, high = spec.action_limits()
            lo, hi = torch.from_numpy(low).cuda(), torch.from_numpy(high).cuda()
            for E in (1024, 4096, 16384, 65536):
                if B * E > 1024 * 16384:
                    continue
                acts = [lo[:, None] + torch.rand((len(low), E), device='cuda') * (hi - lo)[:, None] for _ in range(2)]
                row = {'B': B, 'E': E, 'precision': sys.argv[2] if len(sys.argv) > 2 else 'chain'}
                for label, tun in (('default', {}), ('32', dict(b_chunk=32, nw=16)), ('64', dict(b_chunk=64, nw=16)), ('128', dict(b_chunk=128, nw=16)), ('256', dict(b_chunk=256, nw=16))):
                    if tun and tun['b_chunk'] > B:
                        continue
                    try:
                        eng = StepEngine(tab, E, f64_maps=prec, tuning=dict(finish=3, **tun))
                        eng.trace_kernels()
                        us = measure(eng, acts, steps=40 if B * E >= 2 ** 22 else 100, reps=3)
                        row[label] = round(us, 2)
                        if not tun:
                            row['kernel'] = eng.last_kernels
                        del eng
                    except Exception as e:          # a chunk size the launch refuses (LDS staging, scratch rows)
                        row[label] = str(e)[:80]
                print(json.dumps(row), flush=True)
                f.write(json.dumps(row) + '\n'); f.flush()


if __name__ == '__main__':
    main()
