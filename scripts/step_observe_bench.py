"""step + compact observation: two launches (cl_step_f32, cl_observe_f32) against cl_step_observe_f32 (GPU box)."""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests'))
from golden_util import golden
import os
from citylearn_amd import _lib
if os.environ.get('CL_ALT_LIB'):
    _lib.LIB_PATH = Path(os.environ['CL_ALT_LIB']).resolve()
from citylearn_amd.engine import StepEngine
from citylearn_amd.observations import ObservationLayout
from citylearn_amd.observe import ObservationWriter


def timed(fn, n=50, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (2 * n) * 1e3


# (round 6: thermal districts too -- cl_step_observe_f32 is one launch there as well; precision = the engine's default unless CL_SOB_FP32=1)
F64 = False if os.environ.get('CL_SOB_FP32') else None
for name, E in [('g2022_all', int(x)) for x in sys.argv[1:]] or (('g2022_all', 65536), ('g2022_all', 131072), ('g2022_all', 16384), ('g2020_cz1', 65536), ('g2020_cz1', 16384)):
    spec = golden(name).spec(); tab = spec.episode_tables(0)
    for normalize in (False, True):
        dep_tables, cols = ObservationLayout(spec, 'current', normalize).episode(tab).compact()
        eng = StepEngine(tab, E, f64_maps=F64)
        eng.trace_kernels()
        w = ObservationWriter(eng, dep_tables, None)
        low, high = spec.action_limits()
        lo, hi = torch.from_numpy(low).cuda()[:, None], torch.from_numpy(high).cuda()[:, None]
        acts = (lo + torch.rand((eng.n_act_cols, E), device='cuda') * (hi - lo)).contiguous()
        step = sorted(timed(lambda: eng.step(acts, 7)) for _ in range(3))[1]
        two = sorted(timed(lambda: (eng.step(acts, 7), w.write(8))) for _ in range(3))[1]
        one = sorted(timed(lambda: eng.step_observe(acts, w, 7)) for _ in range(3))[1]
        print(f'{name} {eng.n_bldg} x {E}, {len(cols)} dependent columns, normalised={normalize}: step {step:.2f} us | step + observe {two:.2f} us | '
              f'cl_step_observe_f32 {one:.2f} us  ({eng.n_bldg * E / one * 1e6:.3e} building-timesteps/s with observations)  {eng.last_kernels}', flush=True)
        del eng, w
