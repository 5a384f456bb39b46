"""Per-env-block episode windows (`cl_dims.env_row0`): every block of CL_ROW0_BLOCK envs replays its own window of the
simulation period.  Oracle: the same engine run on tables cut to that window (itself pinned on the reference by the
parity suites) -- results must be bit-identical.  GPU only."""
import numpy as np
import pytest
import torch

from golden_util import golden

pytestmark = pytest.mark.gpu


def _cut(tables, start, n):
    from citylearn_amd.schema import EpisodeTables
    return EpisodeTables(params=tables.params, ts=np.ascontiguousarray(tables.ts[start:start + n]), start=tables.start + start,
                         end=tables.start + start + n - 1, outage=tables.outage[start:start + n])


@pytest.mark.parametrize('name,kind', [('g2022_all', 'MARL'), ('g2020_cz1', 'RewardFunction'), ('g2023_p2', 'SolarPenaltyReward')])
def test_step_and_rollout_with_block_offsets(name, kind):
    from citylearn_amd import abi
    from citylearn_amd.engine import StepEngine
    g = golden(name)
    spec = g.spec()
    tables = spec.episode_tables(0)
    K, offsets = 30, [0, 7, 113, 250]
    E = abi.CL_ROW0_BLOCK * len(offsets)
    gen = torch.Generator(device='cuda').manual_seed(3)
    low, high = (torch.from_numpy(x).cuda() for x in spec.action_limits())
    for detail in (False, True):
        eng = StepEngine(tables, E, reward=kind, detail=detail, n_steps=K, env_row0=offsets)
        refs = [StepEngine(_cut(tables, o, K), abi.CL_ROW0_BLOCK, reward=kind, detail=detail) for o in offsets]
        for t in range(K):
            a = low[:, None] + torch.rand((eng.n_act_cols, E), device='cuda', generator=gen) * (high - low)[:, None]
            eng.step(a, t)
            for j, r in enumerate(refs):
                sl = slice(j * abi.CL_ROW0_BLOCK, (j + 1) * abi.CL_ROW0_BLOCK)
                r.step(a[:, sl].contiguous(), t)
                assert torch.equal(eng.state[:, :, sl], r.state), (t, j)
                assert torch.equal(eng.out_env[:, sl], r.out_env), (t, j)
                planes = range(abi.CL_NO - 1) if detail else (abi.CLO_NET, abi.CLO_REWARD)
                for p in planes:
                    assert torch.equal(eng.out_bldg[p][:, sl], r.out_bldg[p]), (t, j, p)
        with pytest.raises(Exception):
            eng.step(a, K)                                   # t outside the episode
    # fused rollout with the on-device policy: same Philox stream (keyed by env index) -> compare against K single steps
    if name != 'g2020_cz1':
        eng = StepEngine(tables, E, reward=kind, n_steps=K, env_row0=offsets)
        eng.set_action_limits(*spec.action_limits())
        ret = torch.zeros(E, device='cuda')
        eng.rollout(K, seed=11, ret_env=ret)
        one = StepEngine(tables, E, reward=kind, n_steps=K, env_row0=offsets)
        acc = torch.zeros(E, device='cuda')
        lib = one.lib
        import ctypes
        lib.cl_philox_uniform.restype = ctypes.c_float
        lib.cl_philox_uniform.argtypes = [ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32]
        lo, hi = spec.action_limits()
        envs = [0, 255, 256, 700, E - 1]
        for t in range(K):
            a = torch.zeros((one.n_act_cols, E), device='cuda')
            u = np.array([[lib.cl_philox_uniform(11, e, c, t) for e in envs] for c in range(one.n_act_cols)], dtype=np.float32)
            a[:, envs] = torch.from_numpy(lo[:, None] + u * (hi - lo)[:, None]).cuda()
            one.step(a, t)
            acc += one.district_reward
        for p in (abi.CLS_B_SOC, abi.CLS_DS_SOC):
            np.testing.assert_allclose(eng.state[p][:, envs].cpu().numpy(), one.state[p][:, envs].cpu().numpy(), rtol=2e-6, atol=2e-6)
        np.testing.assert_allclose(ret[envs].cpu().numpy(), acc[envs].cpu().numpy(), rtol=1e-5, atol=1e-4)


def test_engine_validates_offsets():
    from citylearn_amd.engine import StepEngine
    g = golden('g2022_all')
    tables = g.spec().episode_tables(0)
    with pytest.raises(ValueError, match='entries'):
        StepEngine(tables, 512, n_steps=10, env_row0=[0])
    with pytest.raises(ValueError, match='inside'):
        StepEngine(tables, 512, n_steps=10, env_row0=[0, tables.n_steps - 9])


@pytest.mark.parametrize('normalize', [False, True])
def test_vector_env_with_episode_offsets_matches_windowed_envs(normalize):
    """2023 schema (LSTM temperature stage, ComfortReward, outage): the observation matrix, rewards and indoor
    temperatures of block g equal those of a plain env whose simulation period starts at that block's offset."""
    from citylearn_amd import abi
    from citylearn_amd.vector_env import VectorCityLearnEnv
    g = golden('g2023_p2')
    K, offsets = 40, [0, 31, 200]
    E = abi.CL_ROW0_BLOCK * len(offsets)
    env = VectorCityLearnEnv(g.schema_path, E, observations='tensor', normalize_observations=normalize, episode_time_steps=K,
                             env_episode_offsets=offsets, simulate_power_outage=False)   # outage draws depend on the window length
    assert env.time_steps == K
    refs = [VectorCityLearnEnv(g.schema_path, abi.CL_ROW0_BLOCK, observations='tensor', normalize_observations=normalize,
                               simulation_start_time_step=o, simulation_end_time_step=o + K - 1,
                               simulate_power_outage=False) for o in offsets]
    obs, _ = env.reset()
    sl = [slice(j * abi.CL_ROW0_BLOCK, (j + 1) * abi.CL_ROW0_BLOCK) for j in range(len(offsets))]
    # observation limits follow the simulation period, which differs between `env` and the windowed references: compare
    # normalised observations through the un-normalised value
    lo_e, hi_e = env.layout.limits()
    def denorm(x, layout):
        if not normalize:
            return x
        lo, hi = layout.limits()
        return x * torch.from_numpy(hi - lo).float().cuda() + torch.from_numpy(lo).float().cuda()
    for j, r in enumerate(refs):
        o_r, _ = r.reset()
        np.testing.assert_allclose(denorm(obs[sl[j]], env.layout).cpu().numpy(), denorm(o_r, r.layout).cpu().numpy(), rtol=2e-5, atol=2e-4)
    gen = torch.Generator(device='cuda').manual_seed(5)
    for t in range(K - 1):
        a = env.sample_actions(gen)
        obs, rew, term, _, _ = env.step(a)
        for j, r in enumerate(refs):
            o_r, rew_r, term_r, _, _ = r.step(a[:, sl[j]].contiguous())
            assert term == term_r
            assert torch.equal(env.stage.indoor_temp[:, sl[j]], r.stage.indoor_temp), (t, j)
            assert torch.equal(rew[sl[j]], rew_r), (t, j)
            np.testing.assert_allclose(denorm(obs[sl[j]], env.layout).cpu().numpy(), denorm(o_r, r.layout).cpu().numpy(),
                                       rtol=2e-5, atol=2e-4, err_msg=f'{t} {j}')
    assert env.terminated


def test_streaming_kpis_with_episode_offsets():
    """`evaluate()` of a batch whose blocks replay different windows == `evaluate()` of windowed envs fed the same actions
    (energy KPIs and the comfort KPIs of the LSTM stage)."""
    from citylearn_amd import abi
    from citylearn_amd.vector_env import VectorCityLearnEnv
    g = golden('g2023_p2')
    K, offsets = 60, [3, 150]
    E = abi.CL_ROW0_BLOCK * len(offsets)
    kw = dict(kpi=True, simulate_power_outage=False, reward_function='citylearn.reward_function.RewardFunction')
    env = VectorCityLearnEnv(g.schema_path, E, episode_time_steps=K, env_episode_offsets=offsets, **kw)
    refs = [VectorCityLearnEnv(g.schema_path, abi.CL_ROW0_BLOCK, simulation_start_time_step=o, simulation_end_time_step=o + K - 1, **kw)
            for o in offsets]
    gen = torch.Generator(device='cuda').manual_seed(9)
    for t in range(K - 1):
        a = env.sample_actions(gen)
        env.step(a)
        for j, r in enumerate(refs):
            r.step(a[:, j * abi.CL_ROW0_BLOCK:(j + 1) * abi.CL_ROW0_BLOCK].contiguous())
    building, district = env.evaluate()
    for j, r in enumerate(refs):
        sl = slice(j * abi.CL_ROW0_BLOCK, (j + 1) * abi.CL_ROW0_BLOCK)
        b_ref, d_ref = r.evaluate()
        assert set(b_ref) == set(building) and set(d_ref) == set(district)
        for k, v in b_ref.items():
            torch.testing.assert_close(building[k][:, sl], v, rtol=1e-6, atol=1e-9, equal_nan=True, msg=k)
        for k, v in d_ref.items():
            torch.testing.assert_close(district[k][sl], v, rtol=1e-6, atol=1e-9, equal_nan=True, msg=k)


@pytest.mark.parametrize('kind', ['RewardFunction', 'MARL'])
@pytest.mark.parametrize('tuning,f64', [(dict(vec=1, lean_variant=2), False), (dict(vec=2, lean_variant=2), False), (dict(vec=4, lean_variant=2), False),
                                        (dict(vec=2, lean_variant=1), False), (dict(envmajor=1), False), (dict(envmajor=1, vec=2), False),
                                        (dict(vec=4, lean_variant=2), 'chain'), (dict(envmajor=1), 'chain')])
def test_env_pitch_changes_nothing_but_the_row_stride(kind, tuning, f64):
    """`cl_dims.env_pitch` (round 5): the building rows of the state / output planes padded beyond n_env -- what keeps a
    2^20-env batch's 4 MiB row stride from aliasing in the memory system.  A pitched engine (516 envs, rows 772 floats apart) and a plain one
    step, reset, roll out and observe bit for bit alike in the lean kernel at every pack width, the general kernel, the env-major kernel and
    under CLD_F64_CHAIN; `state` / `out_bldg` keep their logical shape; the pad entries never reach a result."""
    from citylearn_amd import abi
    from citylearn_amd.engine import StepEngine
    from citylearn_amd.observations import ObservationLayout
    from citylearn_amd.observe import ObservationWriter
    g = golden('g2022_all')
    spec = g.spec()
    tab = spec.episode_tables(0)
    E = 516
    a, b = StepEngine(tab, E, reward=kind, tuning=tuning, f64_maps=f64), StepEngine(tab, E, reward=kind, tuning=tuning, f64_maps=f64, env_pitch=E + 256)
    assert a.env_pitch == E and b.env_pitch == E + 256 and b.dims.env_pitch == E + 256
    assert b.state.shape == a.state.shape == (abi.CL_NS, 17, E) and b.state.stride(1) == E + 256 and b.out_bldg.stride(1) == E + 256
    b._state_store[:, :, E:] = 7.5e8                       # poison the pad entries: nothing may read them into a result
    b.trace_kernels()
    dep_tables, _ = ObservationLayout(spec, 'current', False).episode(tab).compact()
    wa, wb = ObservationWriter(a, dep_tables, None), ObservationWriter(b, dep_tables, None)
    gen = torch.Generator(device='cuda').manual_seed(5)
    for t in range(16):
        act = torch.rand((a.n_act_cols, E), device='cuda', generator=gen) * 2 - 1
        if t % 2:
            a.step(act, t); oa = wa.write(t + 1).clone()
            ob = b.step_observe(act, wb, t)
        else:
            a.step(act, t); b.step(act, t)
            oa, ob = wa.write(t + 1).clone(), wb.write(t + 1)
        assert torch.equal(a.state, b.state) and torch.equal(a.out_bldg[:2], b.out_bldg[:2]) and torch.equal(a.out_env, b.out_env), (t, b.last_kernels)
        assert torch.equal(oa, ob), t
    if tuning.get('vec', 0) != 4:                              # (the fused rollout has no four-envs-per-lane instantiation to force)
        acts = torch.rand((8, a.n_act_cols, E), device='cuda', generator=gen) * 2 - 1
        ra, rb = torch.zeros(E, device='cuda'), torch.zeros(E, device='cuda')
        a.rollout(8, actions=acts, ret_env=ra); b.rollout(8, actions=acts, ret_env=rb)
        assert torch.equal(a.state, b.state) and torch.equal(a.out_env, b.out_env) and torch.equal(ra, rb)
    a.reset(); b.reset()
    assert torch.equal(a.state, b.state)


def test_env_pitch_defaults_and_refusals():
    from citylearn_amd import _lib, abi
    from citylearn_amd.engine import StepEngine
    tab = golden('g2022_all').spec().episode_tables(0)
    big = StepEngine(tab, 524288)
    assert big.env_pitch == 524288 + 256 and big.state.shape[-1] == 524288          # padded by default where the stride would alias
    del big
    assert StepEngine(tab, 65536).env_pitch == 65536 and StepEngine(tab, 262144).env_pitch == 262144
    big = StepEngine(tab, 524288, kpi=True)
    assert big.env_pitch == 524288
    del big
    with pytest.raises(ValueError):
        StepEngine(tab, 512, env_pitch=514)
    with pytest.raises(ValueError):
        StepEngine(golden('g2020_cz1').spec().episode_tables(0), 512, env_pitch=768)     # thermal district: no pitch
    eng = StepEngine(tab, 512)
    eng.dims.env_pitch = 500
    with pytest.raises(_lib.EngineError) as e:
        eng.step(torch.zeros((eng.n_act_cols, 512), device='cuda'))
    assert e.value.code == abi.CL_EINVAL and 'env_pitch' in str(e.value)
