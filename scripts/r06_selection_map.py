"""Re-record tests/golden/kernel_selection_r06.json: the kernel the default launch of every cell of the map selects (GPU box).  A rule change that moves
cells comes with the measurement that justified it (scripts/r06_cliffs.py, scripts/r06_chunk_sweep.py); this script only re-reads the names.
Usage: r06_selection_map.py out.json"""
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests'))
from golden_util import golden
from citylearn_amd.engine import StepEngine
from citylearn_amd.synthetic import tile_district

cells = json.loads((ROOT / 'tests' / 'golden' / 'kernel_selection_r06.json').read_text())
moved = 0
for cell in cells:
    base = golden('g2022_all' if cell['kind'] == 'lean' else 'g2020_cz1').spec()
    B, E = cell['B'], cell['E']
    spec = tile_district(base, B, jitter=0.0 if B <= len(base.buildings) else 0.1)
    eng = StepEngine(spec.episode_tables(0), E, tuning={'finish': 3} if B > 32 else None)
    eng.trace_kernels()
    eng.step(torch.zeros((eng.n_act_cols, E), device='cuda'), 1)
    if eng.last_kernels != cell['kernel']:
        print(f"{cell['kind']} {B} x {E}: {cell['kernel']} -> {eng.last_kernels}")
        cell['kernel'] = eng.last_kernels
        moved += 1
    del eng
Path(sys.argv[1]).write_text(json.dumps(cells, indent=0) + '\n')
print(f'{moved} cells moved')
