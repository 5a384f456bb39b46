"""Upper bound of what balancing the headline's 17th building could buy (GPU box): the lean kernel on 16, 17, 18 and 20 buildings
(2022 device set) x 65 536 envs.  16 buildings = one per wave; 17 .. 20 give 1 .. 4 waves a second building."""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests')); sys.path.insert(0, str(ROOT / 'scripts'))
from golden_util import golden
from citylearn_amd.engine import StepEngine
from citylearn_amd.synthetic import tile_district
from c4_bench import measure
E = 65536
for B in (16, 17, 18, 20, 24, 32):
    spec = golden('g2022_all').spec()
    if B != 17: spec = tile_district(spec, B)
    tab = spec.episode_tables(0)
    eng = StepEngine(tab, E)
    acts = [torch.rand((eng.n_act_cols, E), device='cuda') * 2 - 1 for _ in range(2)]
    us = sorted(measure(eng, acts, steps=60, reps=5) for _ in range(3))[1]
    print(f'{B} x {E}: {us:.2f} us  {us / B * 17:.2f} us scaled to 17 buildings  ({eng.algorithmic_bytes_per_unit() * B * E / us / 1e6:.2f} TB/s)', flush=True)
    del eng, acts
