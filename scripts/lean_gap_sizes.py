"""Lean step between the tuned batch sizes (GPU box): default selection vs the env-major kernel vs the latency-ordered lean kernel."""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests')); sys.path.insert(0, str(ROOT / 'scripts'))
from golden_util import golden
from citylearn_amd.engine import StepEngine
from c4_bench import measure
tab = golden('g2022_all').spec().episode_tables(0)
for E in [int(x) for x in sys.argv[1:]] or (73728, 81920, 98304, 114688, 131072, 163840):
    acts = [torch.rand((17, E), device='cuda') * 2 - 1 for _ in range(2)]
    res = []
    for label, tun in (('default', {}), ('env-major', dict(envmajor=1)), ('lean kernel', dict(envmajor=2, lean_variant=2)), ('general', dict(envmajor=2, lean_variant=1))):
        eng = StepEngine(tab, E, tuning=tun)
        res.append(f'{label} {min(measure(eng, acts, steps=40, reps=4) for _ in range(2)):.2f}')
        del eng
    print(f'17 x {E}: ' + ' | '.join(res) + ' us', flush=True)
