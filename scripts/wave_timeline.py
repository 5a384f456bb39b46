"""Per-wave timeline of cl_step_full_kernel / cl_step_lean_kernel on the GPU box: builds (here or on the CPU container) the diagnostic library with
-DCL_TRACE, replays 3000 back-to-back launches and prints, for the last one, when each phase of each wave happened (REFCLK, 10 ns).

    python scripts/wave_timeline.py build                      # cross-compile citylearn_amd/libcitylearn_amd_trace.so
    python scripts/wave_timeline.py [fixture=g2020_cz1] [envs=65536] [vec=0] [nw=0] [bldgs=0] [b_chunk=0] [nt_stores=0]

Slots per wave: 0 entry (kernel arguments loaded) | per building i of the wave: 1+4i inputs arrived (all loads issued so far
returned), 2+4i arithmetic done, 3+4i stores issued | 13 before the district reduction | 14 everything acknowledged | 15 HW_ID."""
import ctypes
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests'))
from citylearn_amd import _lib

TRACE_LIB = ROOT / 'citylearn_amd' / 'libcitylearn_amd_trace.so'


def build():
    _lib.build_variant(TRACE_LIB, ['-DCL_TRACE'])
    print('built', TRACE_LIB)


def main(argv):
    if argv and argv[0] == 'build':
        return build()
    opt = dict(fixture='g2020_cz1', envs=65536, vec=0, nw=0, bldgs=0, b_chunk=0, detail=0, nt_stores=0)
    for a in argv:
        k, v = a.split('=')
        opt[k] = v if k == 'fixture' else int(v)
    import numpy as np
    import torch
    from golden_util import golden
    _lib.LIB_PATH = TRACE_LIB
    from citylearn_amd.engine import StepEngine
    from citylearn_amd.synthetic import tile_district

    spec = golden(opt['fixture']).spec()
    if opt['bldgs']:
        spec = tile_district(spec, opt['bldgs'])
    tab = spec.episode_tables(0)
    E = opt['envs']
    tun = {k: opt[k] for k in ('vec', 'nw', 'b_chunk', 'nt_stores') if opt[k]}
    eng = StepEngine(tab, E, detail=bool(opt['detail']), tuning=tun)
    lib = _lib.load()
    lib.cl_trace_set.argtypes = [ctypes.c_void_p]
    low, high = spec.action_limits()
    lo, hi = torch.from_numpy(low).cuda(), torch.from_numpy(high).cuda()
    acts = [(lo[:, None] + torch.rand((len(low), E), device='cuda') * (hi - lo)[:, None]).contiguous() for _ in range(2)]
    n_slots = 16
    buf = torch.zeros((1 << 22,), dtype=torch.int64, device='cuda')          # [workgroup][16 waves][16 slots]
    # every launch overwrites the same slots, so after a long back-to-back run (hipGraph replay: clocks up, caches in their
    # steady state) the buffer holds the LAST launch
    assert lib.cl_trace_set(buf.data_ptr()) == 0
    stream = torch.cuda.Stream()
    n_graph, n_replay = 100, 30
    with torch.cuda.stream(stream):
        for t in range(6):
            eng.step(acts[t & 1], t)
        stream.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=stream):
            for t in range(n_graph):
                eng.step(acts[t & 1], 6 + t)
        gr.replay()
        stream.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record(stream)
        for _ in range(n_replay):
            gr.replay()
        ev1.record(stream)
        stream.synchronize()
    us_launch = ev0.elapsed_time(ev1) * 1e3 / (n_graph * n_replay)
    lib.cl_trace_set(None)
    raw = buf.cpu().numpy().reshape(-1, n_slots)
    used = raw[:, 0] != 0
    w = raw[used].astype(np.int64)
    hw = w[:, 15]
    t0 = w[:, 0].min()
    rel = lambda col: (w[:, col] - t0) * 0.01                              # us
    mhz = np.median((w[:, 12] - w[:, 4]) / np.maximum(w[:, 14] - w[:, 0], 1)) * 100.0
    print(f'{opt}: {eng.n_bldg} buildings x {E} envs, {len(w)} waves; {us_launch:.2f} us per launch over {n_graph * n_replay} back-to-back '
          f'launches (traced build); last launch: first entry -> last acknowledged {rel(14).max():.2f} us; shader clock {mhz:.0f} MHz')
    names = {0: 'entry (arguments loaded)', 1: 'b0 inputs arrived', 2: 'b0 arithmetic done', 3: 'b0 stores issued', 5: 'b1 inputs arrived',
             6: 'b1 arithmetic done', 7: 'b1 stores issued', 9: 'b2 inputs arrived', 10: 'b2 arithmetic done', 11: 'b2 stores issued',
             13: 'before district reduction', 14: 'all acknowledged'}
    print(f'{"phase":28s} {"waves":>6s} {"min":>7s} {"p10":>7s} {"median":>7s} {"p90":>7s} {"max":>7s}   (us after the first wave entered)')
    for col, name in names.items():
        m = w[:, col] != 0
        if not m.any():
            continue
        r = (w[m, col] - t0) * 0.01
        print(f'{name:28s} {int(m.sum()):6d} {r.min():7.2f} {np.percentile(r, 10):7.2f} {np.median(r):7.2f} {np.percentile(r, 90):7.2f} {r.max():7.2f}')
    # durations inside a wave
    def span(a, b, label):
        m = (w[:, a] != 0) & (w[:, b] != 0)
        if m.any():
            d = (w[m, b] - w[m, a]) * 0.01
            print(f'  {label:44s} median {np.median(d):6.2f}  p90 {np.percentile(d, 90):6.2f}  max {d.max():6.2f} us')
    span(0, 1, 'entry -> b0 inputs (param round trip + loads)')
    span(1, 2, 'b0 arithmetic (shared SIMD)')
    span(3, 5, 'b0 stores issued -> b1 inputs arrived')
    span(5, 6, 'b1 arithmetic')
    span(13, 14, 'district reduction + last acknowledgement')
    span(0, 14, 'wave lifetime')
    # per-SIMD occupancy: HW_ID bits simd [5:4], cu [11:8], sh [12], se [15:13]; XCC_ID in the upper word
    simd = (hw & 0xffff) >> 4 & 0xfff | ((hw >> 32) & 0xf) << 12
    ids, counts = np.unique(simd, return_counts=True)
    print(f'  distinct (xcc, se, sh, cu, simd): {len(ids)}; waves per SIMD min {counts.min()} median {int(np.median(counts))} max {counts.max()}')
    busy = []
    for s in ids[:: max(1, len(ids) // 64)]:
        m = simd == s
        busy.append(((w[m, 14].max() - w[m, 0].min()) * 0.01, (w[m, 0].min() - t0) * 0.01))
    busy = np.array(busy)
    print(f'  sampled SIMDs: first entry at {busy[:, 1].min():.2f} .. {busy[:, 1].max():.2f} us, occupied for median {np.median(busy[:, 0]):.2f} us (max {busy[:, 0].max():.2f})')


if __name__ == '__main__':
    main(sys.argv[1:])
