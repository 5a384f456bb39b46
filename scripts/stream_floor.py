"""Streaming floor of the step's byte mix at large batches (GPU box): the copy-floor access pattern (3 state planes + 1 action plane in,
3 state planes + net + reward out, 16-byte accesses) at 17 x E, by store / load policy, next to cl_step_f32 at the same size."""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests')); sys.path.insert(0, str(ROOT / 'scripts'))
import os
from citylearn_amd import _lib
if os.environ.get('CL_ALT_LIB'):
    _lib.LIB_PATH = Path(os.environ['CL_ALT_LIB']).resolve()
from launch_gap import run, MODES
from golden_util import golden
from citylearn_amd.engine import StepEngine
from c4_bench import measure
tab = golden('g2022_all').spec().episode_tables(0)
for E in [int(x) for x in sys.argv[1:]] or (262144, 1048576):
    mb = 17 * E * 36 / 1e6
    for mode in (() if os.environ.get('CL_ALT_LIB') else (0, 3, 8, 11)):
        p, a, _ = sorted(run(mode, 17, E, 5, 1024, n=20, reps=5) for _ in range(3))[1]
        print(f'17 x {E} floor kernel, {MODES[mode]}: period {p:.1f} us ({mb / p:.2f} TB/s), waves alive {a:.1f} us', flush=True)
    for nt in (2, 1):
        eng = StepEngine(tab, E, tuning=dict(nt_stores=nt))
        acts = [torch.rand((eng.n_act_cols, E), device='cuda') * 2 - 1 for _ in range(2)]
        us = sorted(measure(eng, acts, steps=20, reps=4) for _ in range(3))[1]
        print(f'17 x {E} cl_step_f32 ({"nt" if nt == 1 else "plain"} stores): {us:.1f} us ({eng.algorithmic_bytes_per_unit() * 17 * E / us / 1e6:.2f} TB/s)', flush=True)
        del eng, acts
        torch.cuda.empty_cache()
