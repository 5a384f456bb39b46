"""HBM-streaming regime of the headline step (17 buildings, working set >> 256 MB Infinity Cache): launch time against the env count
around 2^20 -- i.e. against the byte stride between consecutive building rows of a plane (n_env x 4 B).  GPU box."""
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
for E in (base, base + 256, base + 512, base + 1024, base + 2048, base + 4096, base + 16384, base - 256, base - 4096, 1000000, 1100000, 1200000, 1048576 + 65536):
    eng = StepEngine(tab, E)
    acts = [(torch.rand((eng.n_act_cols, E), device='cuda') * 2 - 1) for _ in range(2)]
    runs = sorted(measure(eng, acts, steps=20, reps=4) for _ in range(3))
    by = eng.n_bldg * E * eng.algorithmic_bytes_per_unit()
    print(f'E={E} (row stride {E * 4} B = 2^20 x 4 {E - base:+d} envs): {runs[0]:.1f} / {runs[1]:.1f} / {runs[2]:.1f} us  '
          f'{by / runs[1] / 1e3:.0f} GB/s = {by / runs[1] / 1e3 / 80:.1f}% of 8 TB/s', flush=True)
    del eng, acts; torch.cuda.empty_cache()
