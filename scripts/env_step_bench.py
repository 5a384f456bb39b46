"""User-level throughput of `VectorCityLearnEnv.step` (Python -> ctypes -> kernels): eager, as hipGraph replays through
`VectorCityLearnEnv.capture()` (one step per graph) and through `capture_rollout(policy, 24)` (24 closed-loop steps per graph, with a
replay policy = bench.py's workload, and with a torch policy that samples uniform actions on the device); 65 536 envs, the three
observation forms.  GPU box; output tracked as profiles/r03*_env_step_bench.log."""
import sys, time
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from citylearn_amd.data import sample_schema
from citylearn_amd.vector_env import VectorCityLearnEnv

E = 65536
for name in ('citylearn_challenge_2022_phase_all_720h', 'citylearn_challenge_2023_phase_2_local_evaluation_720h'):
    for kw in ({}, {'observations': 'compact', 'normalize_observations': True}, {'observations': 'tensor', 'normalize_observations': True}):
        env = VectorCityLearnEnv(sample_schema(name), E, **kw)
        acts = [env.sample_actions() for _ in range(4)]
        n = 300
        for i in range(20):
            env.step(acts[i % 4])
        torch.cuda.synchronize(); env.reset()
        t0 = time.perf_counter()
        for i in range(n):
            env.step(acts[i % 4])
        torch.cuda.synchronize()
        eager = (time.perf_counter() - t0) / n * 1e6
        # captured: the policy writes into one persistent action buffer; first pass captures, second pass replays
        env.reset()
        buf = acts[0].clone()
        cap = env.capture(buf)
        for i in range(n):
            cap.step()
        torch.cuda.synchronize(); env.reset()
        t0 = time.perf_counter()
        for i in range(n):
            buf.copy_(acts[i % 4], non_blocking=True)           # what a policy's output write costs at least
            cap.step()
        torch.cuda.synchronize()
        captured = (time.perf_counter() - t0) / n * 1e6
        env.reset()
        t0 = time.perf_counter()
        for i in range(n):
            cap.step()
        torch.cuda.synchronize()
        bare = (time.perf_counter() - t0) / n * 1e6
        # closed-loop chunks: 24 x (policy, step) per graph launch
        K = 24
        lo, hi = env.action_low[:, None], env.action_high[:, None]
        policies = {'replay': lambda obs, i: acts[i % 4],
                    'uniform': lambda obs, i: lo + (hi - lo) * torch.rand((env.n_act_cols, E), device=env.device)}
        chunked = {}
        for pname, pol in policies.items():
            env.reset()
            roll = env.capture_rollout(pol, K, keep_rewards=env.central_agent)
            n_chunks = (env.time_steps - 1) // K
            for c in range(n_chunks):                               # capture (and run) every chunk of the episode once
                roll.run()
            torch.cuda.synchronize(); env.reset()
            t0 = time.perf_counter()
            for c in range(n_chunks):
                roll.run()
            torch.cuda.synchronize()
            chunked[pname] = (time.perf_counter() - t0) / (n_chunks * K) * 1e6
            del roll
        units = env.n_bldg * E
        print(f'{name.split("_720h")[0]} {kw.get("observations", "planes")}: capture_rollout(k = {K}): replay policy {chunked["replay"]:.2f} us / step '
              f'({units / chunked["replay"] * 1e6:.3e} building-timesteps/s), on-device uniform policy {chunked["uniform"]:.2f} us / step', flush=True)
        print(f'{name.split("_720h")[0]} {kw.get("observations", "planes")}: eager {eager:.1f} us / step ({units / eager * 1e6:.3e} building-timesteps/s)   '
              f'captured {captured:.1f} us incl. a 4.5 MB action copy, {bare:.1f} us without ({units / bare * 1e6:.3e})', flush=True)
        del env, cap
        torch.cuda.empty_cache()
