"""LSTM stage: worst errors vs the reference and kernel timing (GPU box)."""
import sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests'))
import os
from citylearn_amd import _lib
if os.environ.get('CL_ALT_LIB'):           # A/B an alternative build of the library
    _lib.LIB_PATH = Path(os.environ['CL_ALT_LIB']).resolve()
from golden_util import golden
from citylearn_amd.engine import StepEngine
from citylearn_amd.dynamics import LSTMStage
g = golden('g2023_p2'); spec = g.spec(); tab = spec.episode_tables(0); attrs = spec.reward_function['attributes']
E = 64
QUICK = '--quick' in sys.argv     # production variant only: accuracy, then timing at a few batch sizes (+ the ablations)
SPLITS = [a.split('=')[1] for a in sys.argv[1:] if a.startswith('split=')] or ['f16']      # split=f16 split=bf16
cool = torch.from_numpy(g.ref['cool_dem']).cuda()
LABEL = {0: ' split matrix-core path', 1: ' [f32 MFMA, experiment: no activations]', 2: ' [f32 MFMA, experiment: no MFMA]',
         3: ' f32-MFMA path', 8: ' two-term split-bf16 (3 partial products)', 5: ' [split, experiment: no activations]',
         6: ' [split, experiment: no MFMA]', 32: ' [split, experiment: common-denominator cell update, 7 transcendentals per unit and cell]'}
for split in SPLITS:
    for dbg in ((0, 32) if QUICK else ((0, 32, 8, 3) if split == 'bf16' else (0, 32))):
        eng = StepEngine(tab, E, detail=True, tuning=dict(lstm_variant=dbg))
        stage = LSTMStage(spec, tab, eng, attrs['band'], attrs['lower_exponent'], attrs['higher_exponent'], split=split, cell_update='plain')
        wt = wr = 0.0
        for t in range(g.facts['steps']):
            temp = stage.step(t, cool[t][:, None].expand(-1, E).contiguous())
            tt, rr = temp.cpu().numpy(), stage.comfort.cpu().numpy()
            wt = max(wt, float(np.max(np.abs(tt[:, 0] - g.ref['indoor_temp'][t]))))
            ref = g.ref['reward_ComfortReward'][t]
            wr = max(wr, float(np.max(np.abs(rr[:, 0] - ref) / (1e-4 + 1e-4 * np.abs(ref)))))
        print(f'{split} variant {dbg}: teacher-fed worst |dT| =', wt, 'C ; worst comfort reward err / (1e-4 + 1e-4|ref|) =', wr, flush=True)
    if QUICK:
        cases = ((4096, 0), (16384, 0), (65536, 0), (262144, 0), (65536, 32), (4096, 32), (65536, 5), (65536, 6))
    else:
        cases = ((4096, 0), (65536, 0), (65536, 32), (65536, 5), (65536, 6)) + (((4096, 8), (65536, 8), (4096, 3), (65536, 3), (65536, 1), (65536, 2)) if split == 'bf16' else ())
    for E_, dbg in cases:
        eng = StepEngine(tab, E_, detail=True, tuning=dict(lstm_variant=dbg))
        stage = LSTMStage(spec, tab, eng, 1.0, 2.0, 3.0, split=split, cell_update='plain')
        cd = torch.rand((3, E_), device='cuda') * 5
        for t in range(12, 16): stage.step(t, cd)
        torch.cuda.synchronize()
        ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
        ev0.record()
        n = 20
        for t in range(20, 20 + n): stage.step(t, cd)
        ev1.record(); torch.cuda.synchronize()
        us = ev0.elapsed_time(ev1) / n * 1e3
        flop = 3 * E_ * 12 * (64 * 18 + 64 * 32) * 2
        print(f'{split} E={E_}{LABEL[dbg]}: {us:.1f} us per LSTM step  {3*E_/us*1e6:.3e} building-timesteps/s  {flop/us/1e6:.1f} TFLOP/s fp32', flush=True)
        del eng, stage, cd
