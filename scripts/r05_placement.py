"""Round 5: does the run-to-run spread of the HBM-streaming shape (17 x 1 048 576: 115 - 129 us between processes) follow the ADDRESSES of the planes?
Twelve engines in one process, each behind a differently sized spacer allocation; per engine: base addresses of the state / output / action storage
(mod 4 MiB, mod 64 MiB, in 2 MiB units) and the step time.  GPU box."""
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
E = 1 << 20
MiB = 1 << 20
keep = []
for trial in range(14):
    spacer = torch.empty(((3 + 5 * trial) % 23 + 1) * MiB // 4 * 2, device='cuda')           # 2 .. 46 MiB, kept alive: shifts what follows
    eng = StepEngine(tab, E)
    acts = [(torch.rand((eng.n_act_cols, E), device='cuda') * 2 - 1) for _ in range(2)]
    runs = sorted(measure(eng, acts, steps=20, reps=4) for _ in range(3))
    ptrs = dict(state=eng._state_store.data_ptr(), out=eng._out_store.data_ptr(), env=eng._out_env.data_ptr(), a0=acts[0].data_ptr(), a1=acts[1].data_ptr())
    desc = ' '.join(f'{k}:{(v // (2 * MiB)) % 32:2d}/32' for k, v in ptrs.items())
    print(f'trial {trial:2d}: {runs[0]:.1f} / {runs[1]:.1f} / {runs[2]:.1f} us   2-MiB slot mod 32 -> {desc}   out-state {((ptrs["out"] - ptrs["state"]) // (2 * MiB)) % 32:2d}  a0-state {((ptrs["a0"] - ptrs["state"]) // (2 * MiB)) % 32:2d}', flush=True)
    keep.append(spacer)
    del eng, acts
    torch.cuda.empty_cache()
