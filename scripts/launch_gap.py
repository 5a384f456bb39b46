"""Where a back-to-back launch period goes (GPU box): the copy-floor access pattern of the headline step (17 x 65 536, 3 state
planes + 1 action plane in, up to 5 planes out) launched 100 x per hipGraph, each wave stamping REFCLK at entry and after its last
store was acknowledged.  period = events / launches; alive = last acknowledgement - first entry of the LAST launch;
gap = period - alive = command processor + end-of-kernel cache maintenance + dispatch ramp.
Store policies: plain (write-back L2), sc1 (write-through, agent scope), sc0 sc1 (system scope), nt (streaming hint)."""
import ctypes, sys
from pathlib import Path
import numpy as np
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from citylearn_amd import _lib
lib = _lib.load_tune()
vp = ctypes.c_void_p
lib.cl_tune_launch_gap.argtypes = [ctypes.c_int, vp, vp, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp]
MODES = {0: 'plain', 1: 'sc1', 2: 'sc0 sc1', 3: 'nt', 4: 'sc0 nt', 5: 'sc1 nt', 6: 'sc0 sc1 nt', 7: 'sc0'}
MODES.update({k + 8: v + ' +ntload' for k, v in list(MODES.items())})


def run(mode, B, E, planes_out, threads, n=100, reps=20):
    st_in = torch.rand((3, B, E), device='cuda'); act = torch.rand((B, E), device='cuda')
    out2 = torch.empty((2, B, E), device='cuda')
    n_waves = (E // 256) * (threads // 64)
    stamps = torch.zeros((n_waves, 2), dtype=torch.int64, device='cuda')
    stream = torch.cuda.Stream()
    fn = lambda: lib.cl_tune_launch_gap(mode, st_in.data_ptr(), act.data_ptr(), st_in.data_ptr(), out2.data_ptr(), B, E, planes_out, threads,
                                        stamps.data_ptr(), torch.cuda.current_stream().cuda_stream)
    with torch.cuda.stream(stream):
        for _ in range(3): assert fn() == 0
        stream.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            for _ in range(n): fn()
        g.replay(); stream.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps): g.replay()
        e1.record(stream); stream.synchronize()
    period = e0.elapsed_time(e1) / (n * reps) * 1e3
    s = stamps.cpu().numpy()
    alive = (s[:, 1].max() - s[:, 0].min()) * 0.01
    ramp = (s[:, 0].max() - s[:, 0].min()) * 0.01
    return period, alive, ramp


if __name__ == '__main__':
    E = 65536
    print('17 x 65536, 1024-thread workgroups; bytes in 4 planes (17.8 MB), out `planes` x 4.46 MB')
    print(f'{"stores":>16s} {"planes out":>10s} {"period us":>10s} {"waves alive":>11s} {"gap":>6s} {"entry ramp":>10s}')
    for planes in (5, 0):
        for mode in ((0, 3, 4, 5, 6, 7, 8, 11, 12) if planes else (0, 8)):
            r = [run(mode, 17, E, planes, 1024) for _ in range(3)]
            p, a, ramp = sorted(r)[1]
            print(f'{MODES[mode]:>16s} {planes:10d} {p:10.2f} {a:11.2f} {p - a:6.2f} {ramp:10.2f}', flush=True)
    print('empty kernel (no buildings): the bare launch period by grid size')
    for e, th in ((65536, 1024), (65536, 256), (65536, 64), (16384, 1024), (256, 64)):
        p, a, ramp = run(0, 0, e, 0, th)
        print(f'  {e // 256:5d} workgroups x {th:4d} threads: period {p:.2f} us, alive {a:.2f}, entry ramp {ramp:.2f}', flush=True)
