"""Per-wave timeline of cl_lstm_kernel on the GPU box (diagnostic library built with -DCL_TRACE, see scripts/wave_timeline.py build):
REFCLK (10 ns) stamps of every window step of every wave of the LAST of a run of back-to-back LSTM steps.

    python scripts/wave_timeline.py build            # cross-compile citylearn_amd/libcitylearn_amd_trace.so (CPU container)
    python scripts/lstm_timeline.py [split=f16|bf16] [envs=65536] [envs=4096] ...

Slots per wave (lane k of the stamp register): 0 entry | 1 weights + carried state + first inputs arrived, first layer-0 gates done |
per window step s: 2+4s W_hh1 h1 done, 3+4s layer-0 cell update + split done, 4+4s layer-1 gates complete (W_ih1 h0), 5+4s layer-1
cell update done | 50+s the inputs fetched at the top of step s have arrived (s < 11) | 62 everything acknowledged | 63 HW_ID."""
import ctypes
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests'))
from citylearn_amd import _lib

TRACE_LIB = ROOT / 'citylearn_amd' / 'libcitylearn_amd_trace.so'


def run(E, split):
    import numpy as np
    import torch
    from golden_util import golden
    from citylearn_amd.engine import StepEngine
    from citylearn_amd.dynamics import LSTMStage

    g = golden('g2023_p2'); spec = g.spec(); tab = spec.episode_tables(0)
    eng = StepEngine(tab, E, detail=True)
    stage = LSTMStage(spec, tab, eng, 1.0, 2.0, 3.0, split=split)
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
    t0 = w[:, 0].min()
    d = lambda a, b: ((w[:, b] - w[:, a]) & 0xffffffff) * 0.01            # us
    print(f'--- {split} split, {B} buildings x {E} envs: {len(w)} waves, {us:.1f} us per LSTM step (traced build); last launch: first entry -> last '
          f'acknowledged {(((w[:, 62] - t0) & 0xffffffff) * 0.01).max():.1f} us')
    med = lambda x: f'median {np.median(x):6.2f}  p10 {np.percentile(x, 10):6.2f}  p90 {np.percentile(x, 90):6.2f}  max {x.max():6.2f} us'
    print(f'  entry -> weights / state / first gates        {med(d(0, 1))}')
    per = {k: [] for k in ('a', 'b', 'c', 'e', 'f')}
    for s in range(12):
        prev = 1 if s == 0 else 5 + 4 * (s - 1)
        per['a'].append(d(prev, 2 + 4 * s)); per['b'].append(d(2 + 4 * s, 3 + 4 * s)); per['c'].append(d(3 + 4 * s, 4 + 4 * s))
        per['e'].append(d(4 + 4 * s, 5 + 4 * s))
        if s < 11:
            per['f'].append(d(5 + 4 * s, 50 + s))
    lab = {'a': 'step top -> W_hh1 h1 done (+ most of cell 0)  ', 'b': '-> layer-0 cell update + split done           ',
           'c': '-> layer-1 gates complete (W_ih1 h0)          ', 'e': '-> layer-1 cell update done (+ next layer 0)  ',
           'f': '-> prefetched inputs arrived (end of step)    '}
    for k in 'abcef':
        x = np.concatenate(per[k])
        print(f'  {lab[k]}{med(x)}')
    step = np.concatenate([d(1 if s == 0 else 5 + 4 * (s - 1), 5 + 4 * s) for s in range(12)])
    print(f'  one window step                               {med(step)}')
    print(f'  window step 0 (waits for the remaining weights){med(d(1, 5))}')
    print(f'  window steps 1..11                            {med(step[len(w):])}')
    print(f'  last cell update -> all acknowledged          {med(d(49, 62))}')
    print(f'  wave lifetime                                 {med(d(0, 62))}')
    hw = w[:, 63]
    simd = ((hw & 0xffff) >> 4 & 0xfff) | (((hw >> 16) & 0xf) << 12)
    ids, counts = np.unique(simd, return_counts=True)
    print(f'  distinct (xcc, se, sh, cu, simd): {len(ids)}; waves per SIMD min {counts.min()} median {int(np.median(counts))} max {counts.max()}')
    # co-residency: for a sample of SIMDs, how many waves were alive at the same time on average
    live = []
    for sid in ids[:: max(1, len(ids) // 128)]:
        m = simd == sid
        a, b = (w[m, 0] - t0) & 0xffffffff, (w[m, 62] - t0) & 0xffffffff
        span = b.max() - a.min()
        live.append(((b - a).sum() / max(span, 1), span * 0.01))
    live = np.array(live)
    print(f'  sampled SIMDs: mean resident waves {live[:, 0].mean():.2f}, occupied for median {np.median(live[:, 1]):.1f} us')


if __name__ == '__main__':
    _lib.LIB_PATH = TRACE_LIB
    sizes = [int(a.split('=')[1]) for a in sys.argv[1:] if a.startswith('envs=')] or [65536, 4096]
    split = ([a.split('=')[1] for a in sys.argv[1:] if a.startswith('split=')] or ['f16'])[0]
    for E in sizes:
        run(E, split)
