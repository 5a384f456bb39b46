"""A/B timing of alternative builds of the library on the GPU box: CL_ALT_LIB=<path> python scripts/alt_lib_time.py [shape ...]
shapes: thermal (2020 9 x 65536), c3 (2023 3 x 65536), lean (2022 17 x 65536); CL_TUNING="vec=2,nw=9" sets cl_tuning fields."""
import os, sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests')); sys.path.insert(0, str(ROOT / 'scripts'))
from citylearn_amd import _lib
if os.environ.get('CL_ALT_LIB'):
    _lib.LIB_PATH = Path(os.environ['CL_ALT_LIB']).resolve()
from golden_util import golden
from citylearn_amd.engine import StepEngine
from c4_bench import measure
SHAPES = {'thermal': ('g2020_cz1', 65536), 'c3': ('g2023_p2', 65536), 'lean': ('g2022_all', 65536), 'lean1m': ('g2022_all', 1048576),
          'c4': ('g2020_cz1', 1024, 1024), 'c4lean': ('g2022_all', 1024, 1024), 'thermal256k': ('g2020_cz1', 262144), 'p3': ('s_2023_p3', 65536)}
for name in (sys.argv[1:] or ['thermal', 'c3', 'lean']):
    fx, E, *tile = SHAPES[name]
    spec = golden(fx).spec()
    if tile:
        from citylearn_amd.synthetic import tile_district
        spec = tile_district(spec, tile[0])
    tab = spec.episode_tables(0)
    low, high = spec.action_limits()
    lo, hi = torch.from_numpy(low).cuda(), torch.from_numpy(high).cuda()
    acts = [lo[:, None] + torch.rand((len(low), E), device='cuda') * (hi - lo)[:, None] for _ in range(2)]
    tun = dict((k, int(v)) for k, v in (kv.split('=') for kv in os.environ.get('CL_TUNING', '').split(',') if kv))
    eng = StepEngine(tab, E, tuning=tun)
    us = sorted(measure(eng, acts, steps=60 if E < 1000000 else 20, reps=5) for _ in range(3))
    print(f'{_lib.LIB_PATH.name} {tun} {name} {eng.n_bldg} x {E}: {us[1]:.2f} us (runs {", ".join(f"{u:.2f}" for u in us)})', flush=True)
    del eng, acts
