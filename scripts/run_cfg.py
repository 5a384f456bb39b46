"""Run N eager steps of one configuration (for rocprofv3): python scripts/run_cfg.py <fixture> <E> [tile_B] [steps]"""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests'))
from golden_util import golden
from citylearn_amd.engine import StepEngine
from citylearn_amd.synthetic import tile_district
fixture, E = sys.argv[1], int(sys.argv[2])
B = int(sys.argv[3]) if len(sys.argv) > 3 else 0
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 100
spec = golden(fixture).spec()
if B: spec = tile_district(spec, B)
tab = spec.episode_tables(0)
eng = StepEngine(tab, E)
low, high = spec.action_limits()
lo, hi = torch.from_numpy(low).cuda(), torch.from_numpy(high).cuda()
acts = [lo[:, None] + torch.rand((len(low), E), device='cuda') * (hi - lo)[:, None] for _ in range(2)]
for t in range(steps): eng.step(acts[t % 2], 1 + t % 600)
torch.cuda.synchronize()
print('done', eng.n_bldg, E, eng.lean)
