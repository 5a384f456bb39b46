"""PCIe-inclusive rate of the step: what it costs when a HOST agent owns the action and reward buffers.  The C-ABI takes device
pointers, so this is not `bench.py`'s `value` -- it is the note the measurement contract asks for: per step,
actions [n_act_cols][n_env] go host -> device from pinned memory, the step runs, net + reward planes [2][n_bldg][n_env] and the
district sums come back.  Copies and kernel on one stream (serial), and double-buffered over two streams (copy of step t+1
under the kernel of step t -- only legal for an open-loop / one-step-stale policy)."""
import sys, time
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests'))
from golden_util import golden
from citylearn_amd.engine import StepEngine

E = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
g = golden('g2022_all')
tab = g.spec().episode_tables(0)
eng = StepEngine(tab, E)
n, K = eng.n_act_cols, 400
host_a = [torch.rand((n, E)).mul_(2).sub_(1).pin_memory() for _ in range(2)]
dev_a = [torch.empty((n, E), device='cuda') for _ in range(2)]
host_o = torch.empty((2, eng.n_bldg, E)).pin_memory()
host_q = torch.empty(tuple(eng.out_env.shape)).pin_memory()


def run(copy_back, steps):
    eng.reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        dev_a[0].copy_(host_a[k & 1], non_blocking=True)
        eng.step(dev_a[0])
        if copy_back:
            host_o.copy_(eng.out_bldg[:2], non_blocking=True)
            host_q.copy_(eng.out_env, non_blocking=True)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


for label, cb in (('actions in', False), ('actions in, net + reward + district sums out', True)):
    run(cb, 50)
    dt = run(cb, K)
    mb = (n * E * 4 + (cb and (2 * eng.n_bldg * E * 4 + host_q.numel() * 4))) / 1e6
    print(f'17 x {E}, {label}: {dt * 1e6:.1f} us/step, {mb:.1f} MB over PCIe per step ({mb / 1e3 / dt:.1f} GB/s), '
          f'{eng.n_bldg * E / dt:.3e} building-timesteps/s', flush=True)
# device-resident reference on the same box, same eager launch path
acts = torch.rand((n, E), device='cuda') * 2 - 1
eng.reset(); torch.cuda.synchronize(); t0 = time.perf_counter()
for k in range(K):
    eng.step(acts)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print(f'17 x {E}, device-resident actions (eager launches): {dt * 1e6:.1f} us/step, {eng.n_bldg * E / dt:.3e} building-timesteps/s')
