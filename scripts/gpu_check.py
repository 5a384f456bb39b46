"""GPU check: parity of cl_step_f32 against the golden reference trajectories (teacher-forced and free-running)
+ a quick timing sweep.  Run on the GPU box:  python scripts/gpu_check.py [parity|timing]
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

STATE_KEYS = (('soc', abi.CLS_B_SOC), ('eff', abi.CLS_B_EFF), ('degcap', abi.CLS_B_DEGCAP), ('cs_soc', abi.CLS_CS_SOC),
              ('hs_soc', abi.CLS_HS_SOC), ('ds_soc', abi.CLS_DS_SOC))


def parity(name, vec, detail, teach, kinds=('RewardFunction',), n_steps=None, E=64):
    g = golden(name); spec = g.spec(); tab = spec.episode_tables(0)
    lib = _lib.load(); lib.cl_debug_set_vec(vec)
    errs = {}
    for kind in kinds:
        eng = StepEngine(tab, E, reward=kind, detail=detail)
        K = g.facts['steps'] if n_steps is None else min(n_steps, g.facts['steps'])
        acts = torch.from_numpy(g.ref['actions']).cuda()
        ref_state = {k: torch.from_numpy(g.ref[k]).cuda() for k, _ in STATE_KEYS}
        for t in range(K):
            if teach and t > 0:
                for k, pl in STATE_KEYS:
                    eng.state[pl] = ref_state[k][t - 1][:, None]
            eng.step(acts[t][:, None].expand(-1, E).contiguous())
            st = eng.state.cpu().numpy(); ob = eng.out_bldg.cpu().numpy(); oe = eng.out_env.cpu().numpy()
            assert (st[:, :, :1] == st).all() and (ob[:, :, :1] == ob).all(), 'envs with equal actions diverged'
            pairs = {k: st[pl, :, 0] for k, pl in STATE_KEYS}
            pairs['net'] = ob[abi.CLO_NET, :, 0]
            if detail:
                pairs.update({'eb': ob[abi.CLO_B_EB, :, 0], 'cool_dem': ob[abi.CLO_COOL_DEM, :, 0], 'c_cool': ob[abi.CLO_C_COOL, :, 0],
                              'c_dhw': ob[abi.CLO_C_DHW, :, 0], 'c_ns': ob[abi.CLO_C_NSL, :, 0]})
            for k, v in pairs.items():
                ref = g.ref[k][t].astype(np.float64)
                e = np.abs(v - ref) / (1e-4 + 1e-4 * np.abs(ref))    # in units of (atol=1e-4, rtol=1e-4)
                errs[k] = max(errs.get(k, 0), float(e.max()))
            ref = g.ref['reward_' + kind][t]
            e = np.abs(ob[abi.CLO_REWARD, :, 0] - ref) / (1e-4 + 1e-4 * np.abs(ref)); errs['rw_' + kind] = max(errs.get('rw_' + kind, 0), float(e.max()))
            e = abs(oe[abi.CLQ_REWARD, 0] - ref.sum()) / (1e-4 + 1e-4 * abs(ref.sum())); errs['drw_' + kind] = max(errs.get('drw_' + kind, 0), float(e))
            for k, q in (('d_net', abi.CLQ_NET), ('d_cost', abi.CLQ_COST), ('d_emission', abi.CLQ_EMISSION)):
                ref = float(g.ref[k][t]); e = abs(oe[q, 0] - ref) / (1e-4 + 1e-4 * abs(ref)); errs[k] = max(errs.get(k, 0), e)
    worst = max(errs.values())
    print(f'{name} vec={vec} {"full" if (detail or not eng.lean) else "lean"} {"teacher-forced" if teach else "free-running"}: worst err/tol {worst:.3f}',
          {k: round(float(v), 3) for k, v in errs.items() if v > 0.2})
    return worst


def timing():
    g = golden('g2022_all'); spec = g.spec(); tab = spec.episode_tables(0)
    lib = _lib.load()
    for E in (4096, 65536, 262144, 1048576):
        for vec in (1, 2, 4):
            lib.cl_debug_set_vec(vec)
            eng = StepEngine(tab, E)
            acts = [(torch.rand((eng.n_act_cols, E), device='cuda') * 2 - 1) for _ in range(4)]
            for t in range(20): eng.step(acts[t % 4], t % 700)
            torch.cuda.synchronize()
            ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
            K = 300
            ev0.record()
            for t in range(K): eng.step(acts[t % 4], 1 + t % 700)
            ev1.record(); torch.cuda.synchronize()
            ms = ev0.elapsed_time(ev1) / K
            units = E * eng.n_bldg
            bpu = eng.algorithmic_bytes_per_unit()
            print(f'E={E} vec={vec}: {ms*1e3:.1f} us/step  {units/ms*1e3:.3e} bts/s  {units*bpu/ms/1e6:.1f} GB/s (alg {bpu:.1f} B/unit)')


if __name__ == '__main__':
    print(torch.cuda.get_device_name(0))
    what = sys.argv[1] if len(sys.argv) > 1 else 'all'
    if what in ('all', 'parity'):
        all_kinds = ('RewardFunction', 'MARL', 'IndependentSACReward', 'SolarPenaltyReward')
        w = 0
        for vec in (1, 2, 4):
            w = max(w, parity('g2022_all', vec, False, True, all_kinds, 200))
            w = max(w, parity('g2022_all', vec, True, True, ('RewardFunction',), 100))
        for name in ('g2020_cz1', 'g2023_p2'):
            w = max(w, parity(name, 1, True, True, all_kinds))
            w = max(w, parity(name, 2, False, True, ('RewardFunction',), 100))
        print('WORST teacher-forced err/tol', w)
        for name in ('g2022_all', 'g2020_cz1', 'g2023_p2'):
            parity(name, 1, False, False)
    if what in ('all', 'timing'):
        timing()
