"""First-contact GPU check: parity of cl_step_f32 against the golden reference trajectories + a quick timing.
Run on the GPU box:  python scripts/gpu_check.py
"""
import sys, time, json
from pathlib import Path
import numpy as np
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests'))
from golden_util import golden
from citylearn_amd import _lib, abi
from citylearn_amd.engine import StepEngine

def parity(name, vec, n_steps=None):
    g = golden(name)
    spec = g.spec()
    tab = spec.episode_tables(0)
    lib = _lib.load(); lib.cl_debug_set_vec(vec)
    E = 256
    errs = {}
    for kind in ('RewardFunction', 'MARL', 'IndependentSACReward', 'SolarPenaltyReward'):
        eng = StepEngine(tab, E, reward=kind, detail=True)
        K = g.facts['steps'] if n_steps is None else n_steps
        acts = torch.from_numpy(g.ref['actions']).cuda()
        for t in range(K):
            a = acts[t][:, None].expand(-1, E).contiguous()
            eng.step(a)
            torch.cuda.synchronize()
            st = eng.state.cpu().numpy(); ob = eng.out_bldg.cpu().numpy(); oe = eng.out_env.cpu().numpy()
            assert (st[:, :, :1] == st).all() and (ob[:, :, :1] == ob).all(), 'envs with equal actions diverged'
            pairs = {'soc': st[abi.CLS_B_SOC, :, 0], 'eff': st[abi.CLS_B_EFF, :, 0], 'degcap': st[abi.CLS_B_DEGCAP, :, 0],
                     'cs_soc': st[abi.CLS_CS_SOC, :, 0], 'ds_soc': st[abi.CLS_DS_SOC, :, 0], 'net': ob[abi.CLO_NET, :, 0],
                     'eb': ob[abi.CLO_B_EB, :, 0], 'cool_dem': ob[abi.CLO_COOL_DEM, :, 0], 'c_cool': ob[abi.CLO_C_COOL, :, 0],
                     'c_dhw': ob[abi.CLO_C_DHW, :, 0], 'c_ns': ob[abi.CLO_C_NSL, :, 0]}
            for k, v in pairs.items():
                ref = g.ref[k][t].astype(np.float64)
                e = np.abs(v - ref) / (1e-4 + 1e-4 * np.abs(ref))    # in units of (atol=1e-4, rtol=1e-4)
                errs[k] = max(errs.get(k, 0), float(e.max()))
            ref = g.ref['reward_' + kind][t]
            e = np.abs(ob[abi.CLO_REWARD, :, 0] - ref) / (1e-4 + 1e-4 * np.abs(ref)); errs['rw_' + kind] = max(errs.get('rw_' + kind, 0), float(e.max()))
            for k, q in (('d_net', abi.CLQ_NET), ('d_cost', abi.CLQ_COST), ('d_emission', abi.CLQ_EMISSION)):
                ref = float(g.ref[k][t]); e = abs(oe[q, 0] - ref) / (1e-4 + 1e-4 * abs(ref)); errs[k] = max(errs.get(k, 0), e)
        if kind != 'RewardFunction':
            K = min(K, 100)
    print(name, 'vec', vec, 'max err / tol:', {k: round(v, 4) for k, v in errs.items()})
    return max(errs.values())

def timing():
    g = golden('g2022_all'); spec = g.spec(); tab = spec.episode_tables(0)
    lib = _lib.load()
    for E in (4096, 65536, 262144):
        for vec in (1, 2, 4):
            lib.cl_debug_set_vec(vec)
            eng = StepEngine(tab, E)
            a = (torch.rand((eng.n_act_cols, E), device='cuda') * 2 - 1)
            for t in range(20): eng.step(a, t % 700)
            torch.cuda.synchronize()
            ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
            K = 200
            ev0.record()
            for t in range(K): eng.step(a, t % 700)
            ev1.record(); torch.cuda.synchronize()
            ms = ev0.elapsed_time(ev1) / K
            units = E * eng.n_bldg
            bpu = eng.algorithmic_bytes_per_unit()
            print(f'E={E} vec={vec}: {ms*1e3:.1f} us/step  {units/ms/1e3:.3e} bts/s  {units*bpu/ms/1e6:.1f} GB/s (alg {bpu:.1f} B/unit)')

if __name__ == '__main__':
    print(torch.cuda.get_device_name(0))
    worst = 0
    for name in ('g2022_all', 'g2020_cz1', 'g2023_p2'):
        for vec in (1, 2, 4):
            worst = max(worst, parity(name, vec, 240 if vec > 1 else None))
    print('WORST err/tol', worst)
    timing()
