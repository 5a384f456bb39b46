"""Streaming floor of the headline step: a kernel with cl_step_kernel's launch shape and byte counts but no energy model,
timed like bench.py (hipGraph replay of 100 launches), next to the real step (GPU box)."""
import ctypes, sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests'))
from golden_util import golden
from citylearn_amd import _lib
from citylearn_amd.engine import StepEngine
lib = _lib.load_tune()        # the floor kernel lives in libcitylearn_amd_tune.so (csrc/cl_tune.hip)


def timed(fn, n=100, reps=10):
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        for _ in range(3): fn()
        stream.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            for _ in range(n): fn()
        g.replay(); stream.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps): g.replay()
        e1.record(stream); stream.synchronize()
    return e0.elapsed_time(e1) / (n * reps) * 1e3


if __name__ == "__main__":
  for E in (65536, 262144):
      B = 17
      st_in = torch.rand((3, B, E), device='cuda'); act = torch.rand((B, E), device='cuda')
      out2 = torch.empty((2, B, E), device='cuda')
      # in place on the state planes, like the step
      us_c = timed(lambda: lib.cl_tune_copy_floor(st_in.data_ptr(), act.data_ptr(), st_in.data_ptr(), out2.data_ptr(), B, E,
                                                             torch.cuda.current_stream().cuda_stream))
      tab = golden('g2022_all').spec().episode_tables(0)
      eng = StepEngine(tab, E)
      a = torch.rand((eng.n_act_cols, E), device='cuda') * 2 - 1
      us_s = timed(lambda: eng.step(a, 5))
      eng.tuning.lean_variant = 1; us_g = timed(lambda: eng.step(a, 5)); eng.tuning.lean_variant = 0
      acts = torch.rand((eng.n_act_cols + 3, E), device='cuda')[3:] * 2 - 1          # same data, but not "column b = building b"
      eng2 = StepEngine(tab, E); eng2.dims.flags &= ~16
      us_n = timed(lambda: eng2.step(a, 5))
      print(f'   generic lean kernel: {us_g:.2f} us; latency-ordered kernel without the action-column hint: {us_n:.2f} us')
      mb = B * E * 36 / 1e6
      print(f'17 x {E}: copy-floor kernel {us_c:.2f} us ({mb / us_c * 1e3 / 1e3:.2f} TB/s)   cl_step_f32 {us_s:.2f} us '
            f'({eng.algorithmic_bytes_per_unit() * B * E / us_s / 1e6:.2f} TB/s)   step / floor = {us_s / us_c:.2f}', flush=True)
