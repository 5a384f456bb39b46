"""Thermal / outage step kernels on the GPU box: (1) bit comparison of cl_step_full_kernel (cl_full.h) with the round-1 general
kernel on random inputs, (2) launch-geometry sweep of both at the shapes VERDICT r01 names.
    python scripts/full_sweep.py [compare] [sweep] [quick]"""
import sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests')); sys.path.insert(0, str(ROOT / 'scripts'))
from golden_util import golden
from citylearn_amd import abi, _lib
import os
if os.environ.get('CL_ALT_LIB'):           # A/B experiments: a second build of the library (e.g. other compiler flags)
    _lib.LIB_PATH = Path(os.environ['CL_ALT_LIB']).resolve()
from citylearn_amd.engine import StepEngine
from citylearn_amd.synthetic import tile_district
from c4_bench import measure

what = sys.argv[1:] or ['compare', 'sweep']


def compare():
    for name, kind, detail in (('g2020_cz1', 'RewardFunction', False), ('g2020_cz1', 'SolarPenaltyReward', True), ('g2023_p2', 'MARL', False),
                               ('g2023_p2', 'IndependentSACReward', True), ('s_2023_p3', 'RewardFunction', True), ('s_baeda', 'RewardFunction', True)):
        spec = golden(name).spec(); tab = spec.episode_tables(0)
        E = 516
        old = StepEngine(tab, E, reward=kind, detail=detail, tuning=dict(full_variant=1, vec=1))
        news = {f'vec{v}': StepEngine(tab, E, reward=kind, detail=detail, tuning=dict(vec=v)) for v in (1, 2)}
        low, high = spec.action_limits()
        lo, hi = torch.from_numpy(low).cuda()[:, None], torch.from_numpy(high).cuda()[:, None]
        gen = torch.Generator(device='cuda').manual_seed(3)
        T = min(tab.ts.shape[0] - 1, 300)
        bad = {}
        for t in range(T):
            a = (lo + torch.rand((old.n_act_cols, E), device='cuda', generator=gen) * (hi - lo)).contiguous()
            a[:, 0] = 0.0; a[:, 1] = lo[:, 0]; a[:, 2] = hi[:, 0]
            old.step(a, t)
            for k, e in news.items():
                e.state.copy_(old.state) if False else None
                e.step(a, t)
                for label, x, y in (('state', e.state, old.state), ('net', e.net, old.net), ('reward', e.reward_bldg, old.reward_bldg),
                                    ('out_env', e.out_env, old.out_env)) + ((('detail', e.out_bldg[2:abi.CLO_RESERVED], old.out_bldg[2:abi.CLO_RESERVED]),) if detail else ()):
                    if not torch.equal(x, y):
                        d = (x - y).abs()
                        key = (k, label)
                        n, m = int((d > 0).sum()), float(d.max())
                        if key not in bad: bad[key] = [t, n, m, float((d / (1e-4 + 1e-4 * y.abs())).max())]
                        else: bad[key][1] += n; bad[key][2] = max(bad[key][2], m); bad[key][3] = max(bad[key][3], float((d / (1e-4 + 1e-4 * y.abs())).max()))
                # keep the engines in lock-step so that one differing bit does not snowball
                e.state.copy_(old.state)
        same12 = torch.equal(news['vec1'].out_bldg[:2], news['vec2'].out_bldg[:2])
        print(f'{name} {kind} detail={detail}: {T} steps; vec1 == vec2 on the last step: {same12}; differences vs round-1 kernel '
              f'(first step, elements, max abs, max / (1e-4 + 1e-4 |ref|)): {bad if bad else "none -- bit-identical"}', flush=True)


def sweep(quick=False):
    shapes = (('2020 9 x 65536', 'g2020_cz1', None, 65536, (0, 3, 5, 9)), ('C3 2023 3 x 65536', 'g2023_p2', None, 65536, (0, 3)),
              ('s_2023_p3 6 x 65536', 's_2023_p3', None, 65536, (0, 3, 6)), ('2020 9 x 262144', 'g2020_cz1', None, 262144, (0, 9)),
              ('C3 2023 3 x 262144', 'g2023_p2', None, 262144, (0,)), ('C4 2020 devices 1024 x 1024', 'g2020_cz1', 1024, 1024, (0, 8, 16)))
    for label, fixture, B, E, nws in shapes:
        spec = golden(fixture).spec()
        if B: spec = tile_district(spec, B)
        tab = spec.episode_tables(0)
        low, high = spec.action_limits()
        lo, hi = torch.from_numpy(low).cuda(), torch.from_numpy(high).cuda()
        acts = [lo[:, None] + torch.rand((len(low), E), device='cuda') * (hi - lo)[:, None] for _ in range(2)]
        for variant in ((0,) if quick else (1, 0)):
            for vec in (1, 2):
                for nw in nws:
                    eng = StepEngine(tab, E, tuning=dict(vec=vec, nw=nw, full_variant=variant))
                    try:
                        us = measure(eng, acts, steps=40, reps=4)
                        units = E * eng.n_bldg
                        print(f'{label} {"round-1 kernel" if variant else "cl_step_full_kernel"} vec={vec} nw={nw}: {us:.2f} us  '
                              f'{units * eng.algorithmic_bytes_per_unit() / us / 1e3:.0f} GB/s', flush=True)
                    except Exception as e:
                        print(f'{label} variant={variant} vec={vec} nw={nw}: {type(e).__name__} {str(e)[:80]}', flush=True)
                    del eng


def lean():
    tab = golden('g2022_all').spec().episode_tables(0)
    for E in (65536, 262144, 1048576):
        eng = StepEngine(tab, E)
        acts = [(torch.rand((eng.n_act_cols, E), device='cuda') * 2 - 1) for _ in range(2)]
        us = [measure(eng, acts, steps=60 if E < 1000000 else 20, reps=5) for _ in range(3)]
        print(f'lean 17 x {E}: {min(us):.2f} us (runs: {", ".join(f"{u:.2f}" for u in us)})  lib={_lib.LIB_PATH.name}', flush=True)
        del eng, acts
        torch.cuda.empty_cache()


if 'lean' in what: lean()
if 'compare' in what: compare()
if 'sweep' in what: sweep('quick' in what)
