"""Map the kernel-selection rules of `cl_step_f32` (csrc/cl_kernels.hip step_impl) away from the four district sizes they were tuned at
(VERDICT r05 item 7): us per step of the DEFAULT launch for B buildings x E envs, battery + PV and thermal districts, the kernel it selected,
and the best of a few forced alternatives (envs per lane, env-major / general / multi-tile kernels) -- a cell where an alternative wins by
more than 10 % is a rule to fix.  hipGraph replay, one box, one session.  Usage: r06_cliffs.py out.jsonl [precision: chain | fp32]"""
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'scripts'))
from citylearn_amd import _lib, load_district
from citylearn_amd.data import sample_schema
from citylearn_amd.engine import StepEngine
from citylearn_amd.synthetic import tile_district
from f64_cost import measure

BS = (3, 6, 9, 17, 20, 33, 64, 128, 256, 512, 1024)
ES = (4096, 16384, 65536, 100000, 262144)
ALTS = {'lean': [dict(vec=1), dict(vec=2), dict(vec=4), dict(envmajor=1), dict(envmajor=2, lean_variant=1), dict(envmajor=2, lean_variant=2), dict(lean_variant=16)],
        'thermal': [dict(vec=1), dict(vec=2), dict(full_variant=5), dict(full_variant=3), dict(full_variant=3, vec=2), dict(b_chunk=32), dict(b_chunk=64), dict(b_chunk=128)]}


def main():
    out_path = sys.argv[1]
    prec = {'chain': 'chain', 'fp32': False}[sys.argv[2] if len(sys.argv) > 2 else 'chain']
    bases = {'lean': load_district(sample_schema('citylearn_challenge_2022_phase_all_720h')),
             'thermal': load_district(sample_schema('citylearn_challenge_2020_climate_zone_1_744h'))}
    with open(out_path, 'a') as f:
        for kind, base in bases.items():
            for B in BS:
                spec = tile_district(base, B, jitter=0.0 if B <= len(base.buildings) else 0.1)
                tab = spec.episode_tables(0)
                low, high = spec.action_limits()
                lo, hi = torch.from_numpy(low).cuda(), torch.from_numpy(high).cuda()
                for E in ES:
                    if B * E > 300e6:
                        continue
                    acts = [lo[:, None] + torch.rand((len(low), E), device='cuda') * (hi - lo)[:, None] for _ in range(2)]
                    steps = 100 if B * E < 20e6 else 20

                    def run(tuning):
                        eng = StepEngine(tab, E, f64_maps=prec, tuning={**({'finish': 3} if B > 32 else {}), **tuning})
                        eng.trace_kernels()
                        us = min(measure(eng, acts, steps=steps, reps=3) for _ in range(2))
                        name = eng.last_kernels
                        del eng
                        return us, name
                    try:
                        d_us, d_name = run({})
                    except Exception as e:                       # noqa: BLE001
                        f.write(json.dumps({'kind': kind, 'B': B, 'E': E, 'error': str(e)[:200]}) + '\n'); f.flush()
                        continue
                    alts = []
                    for t in ALTS[kind]:
                        if ('envmajor' in t and t['envmajor'] == 1 and B > 20) or ('full_variant' in t and t['full_variant'] == 5 and B > 32) or \
                           ('b_chunk' in t and (B <= 32 or t['b_chunk'] >= B)) or (t.get('lean_variant') == 16 and B <= 32):
                            continue
                        try:
                            us, name = run(t)
                        except (_lib.EngineError, ValueError, NotImplementedError):
                            continue
                        if name != d_name:
                            alts.append({'tuning': t, 'us': us, 'kernel': name})
                    best = min(alts, key=lambda a: a['us']) if alts else None
                    rec = {'kind': kind, 'B': B, 'E': E, 'precision': 'chain' if prec else 'fp32', 'us': d_us, 'kernel': d_name,
                           'ns_per_kunit': d_us * 1e3 / (B * E / 1000.0), 'best_alternative': best,
                           'alternative_gain': None if best is None else d_us / best['us'], 'alternatives': alts}
                    f.write(json.dumps(rec) + '\n'); f.flush()
                    print(f"{kind} B={B} E={E}: {d_us:.2f} us {d_name}" + ('' if best is None else f"   best alt {best['us']:.2f} us ({d_us / best['us']:.2f} x) {best['tuning']} {best['kernel']}"), flush=True)
                    del acts
                    torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
