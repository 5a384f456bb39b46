"""User-level throughput of `VectorCityLearnEnv.step` (Python -> ctypes -> kernels), eager and under hipGraph replay."""
import sys, time
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests'))
from golden_util import golden
from citylearn_amd.vector_env import VectorCityLearnEnv

for name, kw in (('g2022_all', {}), ('g2022_all', {'observations': 'tensor'}), ('g2023_p2', {}), ('g2023_p2', {'observations': 'tensor'})):
    E = 65536
    env = VectorCityLearnEnv(golden(name).schema_path, E, **kw)
    acts = [env.sample_actions() for _ in range(4)]
    n = 200
    for i in range(20):
        env.step(acts[i % 4])
    torch.cuda.synchronize(); env.reset()
    t0 = time.perf_counter()
    for i in range(n):
        env.step(acts[i % 4])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f'{name} {kw or "planes"}: eager {dt / n * 1e6:.1f} us per env.step  ({env.n_bldg * E * n / dt:.3e} building-timesteps/s)', flush=True)
