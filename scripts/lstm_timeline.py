"""Per-wave timeline of cl_lstm_kernel on the GPU box (diagnostic library built with -DCL_TRACE, see scripts/wave_timeline.py build):
REFCLK (10 ns) stamps of every window step of every wave of the LAST of a run of back-to-back LSTM steps.

    python scripts/wave_timeline.py build            # cross-compile citylearn_amd/libcitylearn_amd_trace.so (CPU container)
    python scripts/lstm_timeline.py [split=f16|bf16] [envs=65536] [envs=4096] ...
    python scripts/lstm_timeline.py load=gpurun_out/lstm_timeline_f16_65536.npy     # re-print a saved run (no GPU)

Slots per wave (lane k of the stamp register; the window loop runs layer 1 one step behind layer 0, csrc/cl_lstm.h):
0 entry | 1 weights + carried state + first inputs arrived, first layer-0 gates done | 3+4s phase A of step s done (layer-0 cell
update + split) | 5+4s layer-1 cell update of step s done (in phase B of step s+1; s = 11: the drain after the loop) | 50+s the inputs
fetched at the top of step s have arrived (end of step s, s < 11) | 62 everything acknowledged | 63 HW_ID | XCC_ID << 16.
The raw stamps of every run are also saved to gpurun_out/lstm_timeline_<split>_<envs>.npy."""
import ctypes
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests'))
from citylearn_amd import _lib

TRACE_LIB = ROOT / 'citylearn_amd' / 'libcitylearn_amd_trace.so'


def capture(E, split):
    import torch
    from golden_util import golden
    from citylearn_amd.engine import StepEngine
    from citylearn_amd.dynamics import LSTMStage

    g = golden('g2023_p2'); spec = g.spec(); tab = spec.episode_tables(0)
    eng = StepEngine(tab, E, detail=True)
    stage = LSTMStage(spec, tab, eng, 1.0, 2.0, 3.0, split=split, cell_update='plain')
    lib = _lib.load()
    lib.cl_trace_set.argtypes = [ctypes.c_void_p]
    B = eng.n_bldg
    n_waves = B * ((E + 127) // 128) * 4
    buf = torch.zeros((n_waves * 64 + 1024,), dtype=torch.int32, device='cuda')
    cd = torch.rand((B, E), device='cuda') * 5
    for t in range(12, 16):
        stage.step(t, cd)
    torch.cuda.synchronize()
    assert lib.cl_trace_set(buf.data_ptr()) == 0
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    ev0.record()
    for t in range(20, 20 + n):
        stage.step(t, cd)
    ev1.record(); torch.cuda.synchronize()
    lib.cl_trace_set(None)
    us = ev0.elapsed_time(ev1) / n * 1e3
    w = buf.cpu().numpy()[: n_waves * 64].reshape(n_waves, 64).astype(np.int64) & 0xffffffff
    w = w[w[:, 0] != 0]
    out = ROOT / 'gpurun_out'
    if out.is_dir():                                   # raw stamps for offline analysis
        np.save(out / f'lstm_timeline_{split}_{E}.npy', w.astype(np.uint32))
    return w, f'{split} split, {B} buildings x {E} envs: {len(w)} waves, {us:.1f} us per LSTM step (traced build)'


def report(w, title):
    w = w.astype(np.int64)
    t0 = w[:, 0].min()
    at = lambda c: ((w[:, c] - t0) & 0xffffffff) * 0.01                   # us after the first wave entered
    d = lambda a, b: ((w[:, b] - w[:, a]) & 0xffffffff) * 0.01            # us
    print(f'--- {title}; last launch: first entry -> last acknowledged {at(62).max():.1f} us')
    med = lambda x: f'median {np.median(x):6.2f}  p10 {np.percentile(x, 10):6.2f}  p90 {np.percentile(x, 90):6.2f}  max {x.max():6.2f} us'
    print(f'  entry -> weights / state / first gates         {med(d(0, 1))}')
    top = lambda s: 1 if s == 0 else 50 + s - 1                                   # where step s starts
    steady = range(2, 11)
    print(f'  phase A (cell 0 of step s || W_hh1 h1)         {med(np.concatenate([d(top(s), 3 + 4 * s) for s in steady]))}')
    print(f'  phase B (cell 1 of step s-1 || W_ih1 h0, L0)   {med(np.concatenate([d(3 + 4 * s, 5 + 4 * (s - 1)) for s in steady]))}')
    print(f'  end of step: prefetched inputs arrived         {med(np.concatenate([d(5 + 4 * (s - 1), 50 + s) for s in steady]))}')
    print(f'  window step 0                                  {med(d(1, 50))}')
    print(f'  window steps 1..10                             {med(np.concatenate([d(50 + s - 1, 50 + s) for s in range(1, 11)]))}')
    print(f'  last step + drain                              {med(d(60, 49))}')
    print(f'  last cell update -> all acknowledged           {med(d(49, 62))}')
    print(f'  wave lifetime                                  {med(d(0, 62))}')
    hw = w[:, 63]
    simd = ((hw & 0xffff) >> 4 & 0xfff) | (((hw >> 16) & 0xf) << 12)              # (xcc, se, sh, cu, simd)
    ids, counts = np.unique(simd, return_counts=True)
    print(f'  distinct (xcc, se, sh, cu, simd): {len(ids)}; waves per SIMD min {counts.min()} median {int(np.median(counts))} max {counts.max()}')
    live = []
    for sid in ids[:: max(1, len(ids) // 128)]:
        m = simd == sid
        a, b = at(0)[m], at(62)[m]
        span = b.max() - a.min()
        live.append(((b - a).sum() / max(span, 1e-9), span))
    live = np.array(live)
    print(f'  sampled SIMDs: mean resident waves {live[:, 0].mean():.2f}, occupied for median {np.median(live[:, 1]):.1f} us')
    # one SIMD, wave by wave: who runs at full speed and who waits (the older wave of a SIMD wins every arbitration)
    sid = ids[np.argmax(counts == counts.max())]          # (the HW_ID / XCC_ID decode is approximate: some keys hold two SIMDs)
    m = np.where(simd == sid)[0]
    print('  one SIMD, its waves in order of arrival (us): entry, first gates, end | duration of window steps 0..10')
    for i in m[np.argsort(at(0)[m])]:
        steps = [d(1, 50)[i]] + [d(50 + s - 1, 50 + s)[i] for s in range(1, 11)]
        print(f'    {at(0)[i]:7.2f} {at(1)[i]:7.2f} {at(62)[i]:7.2f} | ' + ' '.join(f'{x:5.2f}' for x in steps))


if __name__ == '__main__':
    saved = [a.split('=', 1)[1] for a in sys.argv[1:] if a.startswith('load=')]
    if saved:
        for path in saved:
            report(np.load(path), Path(path).name)
        sys.exit(0)
    _lib.LIB_PATH = TRACE_LIB
    sizes = [int(a.split('=')[1]) for a in sys.argv[1:] if a.startswith('envs=')] or [65536, 4096]
    split = ([a.split('=')[1] for a in sys.argv[1:] if a.startswith('split=')] or ['f16'])[0]
    for E in sizes:
        report(*capture(E, split))
