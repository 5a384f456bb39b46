"""Round 5, HBM-streaming headline shape (17 x 1 048 576): cl_step_envmajor_kernel by envs per lane (cl_tuning.vec 1 / 2) and building
bound (17 / 20: lean_variant 8 keeps the general 20), at exactly 2^20 envs and at 2^20 + 256 (row stride off the 4 MiB multiple).  GPU box."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'scripts'))
import torch
from citylearn_amd import load_district
from citylearn_amd.data import sample_schema
from citylearn_amd.engine import StepEngine
from c4_bench import measure
tab = load_district(sample_schema()).episode_tables(0)
base = 1 << 20
variants = [('NB 20, 1 env/lane (round 4)', dict(lean_variant=8)), ('NB 17, 1 env/lane', {}), ('NB 17, 2 envs/lane', dict(vec=2, envmajor=1)),
            ('NB 20, 2 envs/lane', dict(vec=2, envmajor=1, lean_variant=8))]
for E in [int(x) for x in sys.argv[1:]] or (base, base + 256, 262144):
    for label, tun in variants:
        for nt in (0, 2):
            eng = StepEngine(tab, E, tuning=dict(tun, nt_stores=nt))
            eng.trace_kernels()
            acts = [(torch.rand((eng.n_act_cols, E), device='cuda') * 2 - 1) for _ in range(2)]
            runs = sorted(measure(eng, acts, steps=20, reps=4) for _ in range(3))
            by = eng.n_bldg * E * eng.algorithmic_bytes_per_unit()
            print(f'E={E} {label:30s} nt_stores={nt} {eng.last_kernels:50s}: {runs[0]:.1f} / {runs[1]:.1f} / {runs[2]:.1f} us  '
                  f'{by / runs[1] / 1e3:.0f} GB/s = {by / runs[1] / 1e3 / 80:.1f}% of 8 TB/s', flush=True)
            del eng, acts; torch.cuda.empty_cache()
