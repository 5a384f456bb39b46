"""One process = one build of the library: the default launch of 17 x 1 048 576 (and x 2 097 152) under both precision models, median of five 60-launch timings.
Run alternately with CITYLEARN_AMD_LIB pointing at an A/B build (scripts/gpurun/r06_call47.sh)."""
import os
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'scripts'))
from citylearn_amd import load_district
from citylearn_amd.data import sample_schema
from citylearn_amd.engine import StepEngine
from f64_cost import measure

spec = load_district(sample_schema('citylearn_challenge_2022_phase_all_720h'))
tab = spec.episode_tables(0)
low, high = spec.action_limits()
lo, hi = torch.from_numpy(low).cuda(), torch.from_numpy(high).cuda()
tag = Path(os.environ.get('CITYLEARN_AMD_LIB', 'default')).name
for E in (1048576, 2097152):
    acts = [lo[:, None] + torch.rand((len(low), E), device='cuda') * (hi - lo)[:, None] for _ in range(2)]
    for prec, label in ((False, 'fp32'), ('chain', 'chain')):
        eng = StepEngine(tab, E, f64_maps=prec)
        eng.trace_kernels()
        us = sorted(measure(eng, acts, steps=20, reps=3) for _ in range(5))
        print(f'{tag:32s} 17 x {E} {label:5s} median {us[2]:7.2f} us  (min {us[0]:.2f}, max {us[4]:.2f})  {eng.last_kernels}', flush=True)
        del eng
        torch.cuda.empty_cache()
    del acts
    torch.cuda.empty_cache()
