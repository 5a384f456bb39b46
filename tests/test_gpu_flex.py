"""EV chargers + washing machines on the device (`cl_step_flex_f32`, csrc/cl_flex.h) against the oracle
(oracle/flex_oracle.py, pinned to the reference by tests/test_oracle_golden.py) and against the reference's own
trajectory on the 2022 + EVs dataset (tests/golden/g2022_evs)."""
import numpy as np
import pytest

from golden_util import golden

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')


def _drift(g, spec, tab):
    """The N(1, 0.2) multipliers the reference drew for the unconnected-EV SoC drift: its global np.random stream, replayed
    by the oracle under the fixture's seed (the oracle test proves that replay reproduces the reference bit for bit)."""
    from oracle.flex_oracle import FlexDistrictOracle
    np.random.seed(g.facts['seed'])
    if 'noise_seed' in g.facts:          # the loader's `noise_std` draws come out of the same stream first
        spec = g.spec(noise_seed=None)
        tab = spec.episode_tables(0)
    o = FlexDistrictOracle(spec, tab, 1, reward='Electric_Vehicles_Reward_Function')
    o.reset()
    for t in range(g.ref['actions'].shape[0]):
        o.step(g.ref['actions'][t][:, None])
    return o.flex[0].drift_log.astype(np.float32)


@pytest.mark.parametrize('name', ['g2022_evs', 'g_cc_demo', 'g_evs_15min', 'g_evs_central', 'g_evs_noise'])
def test_flex_step_matches_oracle_and_reference(name):
    from citylearn_amd.engine import StepEngine
    from citylearn_amd import abi
    from oracle.flex_oracle import FlexDistrictOracle
    g = golden(name)
    spec = g.spec()
    tab = spec.episode_tables(0)
    drift = _drift(g, spec, tab)
    E = 8
    eng = StepEngine(tab, E, reward='Electric_Vehicles_Reward_Function', detail=True, ev_drift=drift, charger_detail=True,
                     central_agent=spec.central_agent)
    assert eng.flex is not None and eng.n_act_cols == g.ref['actions'].shape[1]
    o = FlexDistrictOracle(spec, tab, 1, reward='Electric_Vehicles_Reward_Function', drift=drift.astype(np.float64))
    o.reset()
    np.testing.assert_allclose(eng.ev_state[0, :, 0].cpu().numpy(), g.ref['ev_soc0'], rtol=1e-6)
    K = g.ref['actions'].shape[0]
    flex_b = tab.flex.flex_bldg
    worst = {}

    plain = {}                                   # the same errors in units of the plain bar, 1e-4 + 1e-4 |ref| (recorded: profiles/r06_parity_worst.md)

    # (round 6: every gate of this test at the plain bar, 1e-4 + 1e-4 |ref| -- measured worst 0.64 x, profiles/r06_parity_worst.md; rounds 3 - 5 gated
    #  the EV planes at 2e-4 and the district sums at 2e-3 absolute)
    def close(name, got, exp, rtol=1e-4, atol=1e-4):
        err = np.abs(got - exp) / (atol + rtol * np.abs(exp))
        worst[name] = max(worst.get(name, 0.0), float(err.max()))
        plain[name] = max(plain.get(name, 0.0), float((np.abs(got - exp) / (1e-4 + 1e-4 * np.abs(exp))).max()))
        np.testing.assert_allclose(got, exp, rtol=rtol, atol=atol, err_msg=f'{name} t={t}')

    flips = 0
    for t in range(K):
        a = torch.from_numpy(np.repeat(g.ref['actions'][t][:, None], E, axis=1)).cuda()
        eng.step(a)
        out = o.step(g.ref['actions'][t][:, None])
        torch.cuda.synchronize()
        ev_soc = eng.ev_state[0].cpu().numpy()
        assert np.all(ev_soc == ev_soc[:, :1])                                # every env saw the same actions
        close('ev_soc', ev_soc[:, 0], out['ev_soc'][:, 0])
        close('ev_soc/ref', ev_soc[:, 0], g.ref['ev_soc'][t])
        close('ev_degcap', eng.ev_state[2, :, 0].cpu().numpy(), g.ref['ev_degcap'][t], rtol=1e-6)
        close('charger_consumption', eng.charger_out[0, :, 0].cpu().numpy(), g.ref['charger_consumption'][t])
        close('charger_energy', eng.charger_out[1, :, 0].cpu().numpy(), g.ref['charger_energy'][t])
        close('chargers_total', eng.flex_out[abi.CLX_CHARGERS, :, 0].cpu().numpy(), g.ref['chargers_total'][t][flex_b])
        close('load', eng.flex_out[abi.CLX_LOAD, :, 0].cpu().numpy(), (g.ref['chargers_total'][t] + g.ref['wms_total'][t])[flex_b])
        close('net', eng.net[:, 0].cpu().numpy(), g.ref['net'][t])
        close('base_net', eng.out_bldg[abi.CLO_BASE_NET, :, 0].cpu().numpy(), g.ref['base_net'][t])
        close('soc', eng.soc[:, 0].cpu().numpy(), g.ref['soc'][t])
        close('d_net', eng.out_env[abi.CLQ_NET, 0].cpu().numpy(), g.ref['d_net'][t])
        close('d_cost', eng.out_env[abi.CLQ_COST, 0].cpu().numpy(), g.ref['d_cost'][t])
        if name == 'g_cc_demo':
            close('violation', eng.flex_out[abi.CLX_VIOLATION, :, 0].cpu().numpy(), g.ref['cc_violation_kwh'][t][flex_b])
            fb = list(flex_b).index(14)                                         # Building_15: limit 12 kW, phases 7 / 5 kW
            head = out['cc_headroom'][14][0]
            got = eng.flex_out[abi.CLX_HEADROOM:abi.CLX_HEADROOM + 3, fb, 0].cpu().numpy()
            close('headroom', got, np.array([head['building'], head['phase_a'], head['phase_b']]))
        # the reward has hard thresholds on SoC differences: allow a float32 / float64 disagreement on a handful of steps
        rw, ref_rw = eng.reward_bldg[:, 0].cpu().numpy(), g.ref['env_rewards'][t]
        if spec.central_agent:                                   # one value: the sum, scaled by the district MARL reward
            rw = eng.out_env[abi.CLQ_REWARD, :1].cpu().numpy()
            np.testing.assert_allclose(rw[0], out['d_reward'][0], rtol=5e-4, atol=5e-4)
        bad = np.abs(rw - ref_rw) > 2e-4 + 2e-4 * np.abs(ref_rw)
        flips += int(bad.sum())
        close('d_reward', eng.out_env[abi.CLQ_REWARD, 0].cpu().numpy(), eng.reward_bldg[:, 0].cpu().numpy().sum(), atol=1e-4)
    assert flips <= 3, flips
    from golden_util import check_worst
    check_worst({k: v for k, v in plain.items() if k not in ('ev_degcap', 'd_reward')}, f'{name} EV district (fp32 batteries)')
    print('worst scaled errors', {k: round(v, 3) for k, v in worst.items()}, 'reward threshold flips', flips)


def test_flex_on_device_drift_is_n_1_02_clipped():
    """Without replayed multipliers the drift comes from Philox: check its distribution on an EV that is away."""
    from citylearn_amd.engine import StepEngine
    from citylearn_amd import abi
    g = golden('g2022_evs')
    spec = g.spec()
    tab = spec.episode_tables(0)
    E = 65536
    eng = StepEngine(tab, E, reward='MARL', ev_seed=1234)
    ev_ts = tab.flex.ev_ts
    # an EV that leaves its charger: held on row t0 - 1 (charged by a positive action), drifting on row t0
    t0, k = next((t, k) for t in range(1, 100) for k in range(ev_ts.shape[1])
                 if ev_ts[t, k, abi.CLEV_RULE_STEP] == -2.0 and ev_ts[t - 1, k, abi.CLEV_CONNECTED] == 1.0)
    a = torch.full((eng.n_act_cols, E), 0.5, device='cuda')
    engines = [eng, StepEngine(tab, E, reward='MARL', ev_seed=1234), StepEngine(tab, E, reward='MARL', ev_seed=99)]
    for t in range(t0 + 1):
        if t == t0:
            before = eng.ev_state[0, k].clone()
        for e in engines:
            e.step(a)
    b = before.cpu().numpy()
    assert np.all(b == b[0]) and 0.05 < b[0] < 0.7            # same actions everywhere: one SoC, away from the clamps
    ratio = (eng.ev_state[0, k] / before).cpu().numpy().astype(np.float64)
    assert 0.6 - 1e-6 <= ratio.min() and ratio.max() <= 1.4 + 1e-6
    assert abs((ratio == ratio.min()).mean() - 0.02275) < 0.004 and abs((ratio == ratio.max()).mean() - 0.02275) < 0.004   # P(|z| > 2)
    inner = ratio[(ratio > 0.6001) & (ratio < 1.3999)]
    assert abs(inner.mean() - 1.0) < 0.005 and abs(inner.std() - 0.2 * 0.8796) < 0.005     # sd of N(0,1) truncated at +-2 is 0.8796
    # the same seed reproduces the stream, another seed does not
    assert torch.equal(engines[1].ev_state[0, k], eng.ev_state[0, k]) and not torch.equal(engines[2].ev_state[0, k], eng.ev_state[0, k])


def _acts(g, env, t):
    a = [float(x) for x in g.ref['actions'][t]]
    out, p = [], 0
    for names in env.action_names:
        out.append(a[p:p + len(names)]); p += len(names)
    return out


@pytest.mark.parametrize('name', ['g2022_evs', 'g_cc_demo', 'g_evs_15min', 'g_evs_central', 'g_evs_noise'])
def test_env_on_the_ev_dataset_matches_the_reference(name):
    """`CityLearnEnv` on the 2022 + EVs schema: names, spaces, the observations reset()/step() return (all 534 columns,
    charger and washing-machine columns included), the Electric_Vehicles_Reward_Function rewards, district series and
    the KPIs of a full episode, against what the reference returned for the same actions (drift multipliers replayed)."""
    from citylearn_amd.citylearn import CityLearnEnv
    g = golden(name)
    spec = g.spec()
    drift = _drift(g, spec, spec.episode_tables(0))
    env = CityLearnEnv(g.schema_path, ev_soc_drift=drift, noise_seed=g.facts.get('noise_seed'))
    for ev, fact in zip(env.district_spec.electric_vehicles, g.facts['electric_vehicles']):      # see golden_util.Golden.spec
        ev.battery.initial_soc = fact['initial_soc']
    assert type(env.reward_function).__name__ == 'Electric_Vehicles_Reward_Function' and env._fused_reward
    assert env.observation_names == g.facts['observation_names'] and env.action_names == g.facts['action_names']
    lo = np.concatenate([s.low for s in env.action_space]); hi = np.concatenate([s.high for s in env.action_space])
    assert np.array_equal(lo, g.ref['action_low']) and np.array_equal(hi, g.ref['action_high'])
    obs, _ = env.reset()
    ref_obs = g.obs['obs']
    np.testing.assert_allclose(np.concatenate(obs), ref_obs[0], rtol=1e-6, atol=1e-6)
    # the NormalizedObservationWrapper view of the same env (charging headroom / violation columns come from the flexible-load
    # planes in every observation mode: the g_cc_demo case)
    from citylearn_amd.wrappers import NormalizedObservationWrapper
    wrapped = NormalizedObservationWrapper(env)
    assert wrapped.observation_names == g.obs_facts['norm_observation_names']
    np.testing.assert_allclose(np.concatenate(wrapped.observation()), g.obs['obs_norm'][0], rtol=1e-6, atol=1e-6)
    K = g.ref['actions'].shape[0]
    flips = 0
    for t in range(K):
        obs, reward, terminated, _, _ = env.step(_acts(g, env, t))
        np.testing.assert_allclose(np.concatenate(obs), ref_obs[t + 1], rtol=1e-6, atol=1e-6 if name == 'g2022_evs' else 5e-4, err_msg=f'obs t={t}')
        if t < 60:
            np.testing.assert_allclose(np.concatenate(wrapped.observation()), g.obs['obs_norm'][t + 1], rtol=1e-5, atol=1e-5 if name == 'g2022_evs' else 5e-4,
                                       err_msg=f'normalised obs t={t}')
        bad = np.abs(np.array(reward) - g.ref['env_rewards'][t]) > 5e-4 + 5e-4 * np.abs(g.ref['env_rewards'][t])
        flips += int(bad.sum())
    assert terminated and flips <= 3, flips
    np.testing.assert_allclose(env.net_electricity_consumption, g.ref['d_net'], rtol=1e-3, atol=3e-3)
    frame = env.evaluate()
    got = {f'{r.level}|{r.name}|{r.cost_function}': r.value for r in frame.itertuples() if r.value is not None and not np.isnan(r.value)}
    ref = dict(zip([str(x) for x in g.ref['kpi_names']], g.ref['kpi_values']))
    n = 0
    for k, v in ref.items():
        if k.split('|')[-1].startswith(('discomfort', 'one_minus_thermal')) or k not in got:
            continue
        if abs(v) > 1e3:
            continue        # ratio over a near-zero baseline sum (15-minute fixture: district baseline emissions ~ 0): ill-conditioned in float32
        np.testing.assert_allclose(got[k], v, rtol=2e-3, atol=2e-4, err_msg=k)
        n += 1
    assert n >= (90 if K > 200 else 60), n


def test_ev_reward_plugin_on_the_host_agrees_with_the_device():
    """A subclass that overrides nothing but is not the stock class takes the host plugin path: its observation
    dictionaries (`electric_vehicles_chargers_dict`, building.py:1340-1389) are rebuilt from the device planes."""
    from citylearn_amd.citylearn import CityLearnEnv
    from citylearn_amd.reward_function import Electric_Vehicles_Reward_Function

    class Mine(Electric_Vehicles_Reward_Function):
        def calculate(self, observations):
            return Electric_Vehicles_Reward_Function.calculate(self, observations)

    g = golden('g2022_evs')
    spec = g.spec()
    drift = _drift(g, spec, spec.episode_tables(0))
    envs = [CityLearnEnv(g.schema_path, ev_soc_drift=drift), CityLearnEnv(g.schema_path, ev_soc_drift=drift, reward_function=Mine)]
    assert envs[0]._fused_reward and not envs[1]._fused_reward
    for env in envs:
        for ev, fact in zip(env.district_spec.electric_vehicles, g.facts['electric_vehicles']):
            ev.battery.initial_soc = fact['initial_soc']
        env.reset()
    flips = 0
    for t in range(60):
        r0 = envs[0].step(_acts(g, envs[0], t))[1]
        r1 = envs[1].step(_acts(g, envs[1], t))[1]
        flips += int((np.abs(np.array(r0) - np.array(r1)) > 1e-3 + 1e-3 * np.abs(np.array(r1))).sum())
        flips += int((np.abs(np.array(r1) - g.ref['env_rewards'][t]) > 2e-3 + 2e-3 * np.abs(g.ref['env_rewards'][t])).sum())
    assert flips <= 2, flips


def test_vector_env_with_evs_and_episode_offsets():
    """Batched form: every env gets its own actions; env 0 replays the fixture.  Then per-env-block episode windows on
    the EV district: each block's EVs start from the reset rule of its own first row."""
    from citylearn_amd.vector_env import VectorCityLearnEnv
    from citylearn_amd import abi
    g = golden('g2022_evs')
    spec = g.spec()
    drift = _drift(g, spec, spec.episode_tables(0))
    E = 512
    env = VectorCityLearnEnv(spec, E, observations='tensor', observation_mode='reference', ev_soc_drift=drift)
    obs, _ = env.reset()
    assert obs.shape == (E, 534) and [n for l in env.observation_names for n in l] == [n for l in g.facts['observation_names'] for n in l]
    np.testing.assert_allclose(obs[0].cpu().numpy(), g.obs['obs'][0], rtol=1e-5, atol=1e-5)
    gen = torch.Generator(device='cuda').manual_seed(5)
    flips = 0
    for t in range(100):
        a = env.sample_actions(gen)
        a[:, 0] = torch.from_numpy(g.ref['actions'][t]).cuda()
        obs, reward, term, _, _ = env.step(a)
        np.testing.assert_allclose(obs[0].cpu().numpy(), g.obs['obs'][t + 1], rtol=1e-5, atol=1e-5)
        assert torch.equal(obs[0], obs[E - 1])                       # reference semantics: every column is exogenous
        r0 = reward[:, 0].cpu().numpy()
        flips += int((np.abs(r0 - g.ref['env_rewards'][t]) > 5e-4 + 5e-4 * np.abs(g.ref['env_rewards'][t])).sum())
        np.testing.assert_allclose(env.engine.ev_state[0, :, 0].cpu().numpy(), g.ref['ev_soc'][t], rtol=2e-4, atol=2e-4)
    assert flips <= 2 and reward.shape == (17, E)
    assert float(env.engine.ev_state[0].std(dim=1).max()) > 0.01         # different actions -> different EV trajectories

    # per-env-block windows of 48 steps over the 240-row simulation period
    off = VectorCityLearnEnv(g.schema_path, E, episode_time_steps=48, env_episode_offsets=[0, 96], observations='planes')
    ft = off.tables.flex
    for blk, row in enumerate((0, 96)):
        rule = ft.ev_ts[row, :, abi.CLEV_RULE_RESET]
        init = np.array([ev.battery.initial_soc for ev in off.district_spec.electric_vehicles], dtype=np.float32)
        expect = np.where(rule >= 0, rule, init)
        got = off.engine.ev_state[0, :, blk * 256:(blk + 1) * 256].cpu().numpy()
        assert np.allclose(got, expect[:, None], atol=1e-6), (blk, got[:, 0], expect)
    a = torch.full((off.n_act_cols, E), 0.3, device='cuda')
    for t in range(47):
        o, r, term, _, _ = off.step(a)
    assert term and torch.isfinite(r).all() and 'electric_vehicle_soc' in o
    # the two blocks replay different charger schedules
    assert not torch.allclose(off.engine.flex_out[abi.CLX_LOAD, :, 0], off.engine.flex_out[abi.CLX_LOAD, :, 300])


def test_observation_tensor_with_charging_constraint_columns():
    """`cl_observe_f32` with CLOB_KIND_EXTRA sources: the headroom / violation columns of the charging-constraints district
    come from the flexible-load planes; everything else from the table.  Env 0 replays the fixture."""
    from citylearn_amd.vector_env import VectorCityLearnEnv
    g = golden('g_cc_demo')
    spec = g.spec()
    drift = _drift(g, spec, spec.episode_tables(0))
    E = 256
    for normalize, key in ((False, 'obs'), (True, 'obs_norm')):
        env = VectorCityLearnEnv(spec, E, observations='tensor', observation_mode='reference', normalize_observations=normalize,
                                 ev_soc_drift=drift)
        obs, _ = env.reset()
        ref = g.obs[key]
        assert obs.shape == (E, ref.shape[1]) and env.writer.n_deps == 4
        np.testing.assert_allclose(obs[0].cpu().numpy(), ref[0], rtol=1e-5, atol=1e-5)
        gen = torch.Generator(device='cuda').manual_seed(11)
        for t in range(60):
            a = env.sample_actions(gen)
            a[:, 0] = torch.from_numpy(g.ref['actions'][t]).cuda()
            obs, reward, _, _, _ = env.step(a)
            np.testing.assert_allclose(obs[0].cpu().numpy(), ref[t + 1], rtol=1e-5, atol=5e-4 if not normalize else 1e-4, err_msg=f't={t}')
        cols = [i for i, n in enumerate([n for l in env.observation_names for n in l]) if 'headroom' in n or 'violation' in n]
        assert len(cols) == 4 and float(obs[:, cols].std(dim=0).max()) > 0          # per-env values


def test_flex_entry_points_validate_their_arguments():
    """Error behaviour of the flexible-load C-ABI: negative codes + message, never a crash."""
    import ctypes
    from citylearn_amd import _lib, abi
    from citylearn_amd.engine import StepEngine
    g = golden('g2022_evs')
    tab = g.spec().episode_tables(0)
    eng = StepEngine(tab, 64, reward='MARL')
    lib = eng.lib
    a = torch.zeros((eng.n_act_cols, 64), device='cuda')
    args = lambda flex, t=0: (ctypes.byref(eng.dims), eng.params.data_ptr(), eng.ts.data_ptr(), eng.state.data_ptr(), a.data_ptr(),
                              a.stride(0), a.stride(1), eng.out_bldg.data_ptr(), eng.out_env.data_ptr(), None, None, flex, t, None)
    good = eng.flex
    assert lib.cl_step_flex_f32(*args(ctypes.byref(good))) == 0
    bad = _lib.Flex.from_buffer_copy(good)
    bad.flex_out = None
    assert lib.cl_step_flex_f32(*args(ctypes.byref(bad))) == abi.CL_ENULL and b'flex_out' in lib.cl_last_error()
    bad = _lib.Flex.from_buffer_copy(good)
    bad.n_rows = 3                                                   # fewer schedule rows than step-table rows
    assert lib.cl_step_flex_f32(*args(ctypes.byref(bad))) == abi.CL_ERANGE
    bad = _lib.Flex.from_buffer_copy(good)
    bad.n_flex_bldg = 0
    assert lib.cl_flex_reset_f32(ctypes.byref(eng.dims), ctypes.byref(bad), None) == abi.CL_EINVAL
    assert lib.cl_flex_reset_f32(ctypes.byref(eng.dims), None, None) == abi.CL_ENULL
    assert lib.cl_step_flex_f32(*args(ctypes.byref(good), t=eng.n_steps)) == abi.CL_ERANGE
    # the EV reward needs the flexible-load tables; the plain entry point refuses it
    ev = StepEngine(tab, 64, reward='Electric_Vehicles_Reward_Function')
    rc = lib.cl_step_f32(ctypes.byref(ev.dims), ev.params.data_ptr(), ev.ts.data_ptr(), ev.state.data_ptr(), a.data_ptr(), a.stride(0), a.stride(1),
                         ev.out_bldg.data_ptr(), ev.out_env.data_ptr(), None, None, 0, None)
    assert rc == abi.CL_EINVAL and b'CLR_EV' in lib.cl_last_error()
    # the K-step launch sequence: same range / pointer checks as cl_rollout_f32
    roll = lambda flex, actions, low, scratch, t0, k: lib.cl_rollout_seq_f32(
        ctypes.byref(eng.dims), eng.params.data_ptr(), eng.ts.data_ptr(), eng.state.data_ptr(), actions, 0, a.stride(0), a.stride(1), low, low, 0,
        scratch, eng.out_bldg.data_ptr(), eng.out_env.data_ptr(), None, None, None, flex, t0, k, None)
    assert roll(ctypes.byref(good), a.data_ptr(), None, None, eng.n_steps - 1, 2) == abi.CL_ERANGE
    assert roll(ctypes.byref(good), None, None, None, 0, 1) == abi.CL_ENULL and b'act_low' in lib.cl_last_error()
    lim = torch.zeros(eng.n_act_cols, device='cuda')
    assert roll(ctypes.byref(good), None, lim.data_ptr(), None, 0, 1) == abi.CL_ENULL and b'policy_actions' in lib.cl_last_error()
    with pytest.raises(ValueError, match='set_action_limits'):
        eng.rollout(4)


@pytest.mark.parametrize('seed', [1, 2, 3, 4])
def test_synthetic_charger_schedules_match_the_oracle(seed, tmp_path):
    """Connection patterns the shipped dataset does not contain (tests/flex_synth.py: EVs swapping chargers, arrivals with and
    without an announced SoC, back-to-back connections, an unused charger), free-running, per-env random actions with exact
    zeros; the oracle is pinned on these very files against the reference by oracle/ref_harness/check_flex_synth.py."""
    from pathlib import Path
    from citylearn_amd.engine import StepEngine
    from citylearn_amd.schema import load_district
    from citylearn_amd import abi
    from oracle.flex_oracle import FlexDistrictOracle
    from flex_synth import make
    g = golden('g2022_evs')
    schema = make(Path(g.schema_path).parent, tmp_path / 'synth', seed, curves=seed % 2 == 0)     # even seeds: charger efficiency curves
    spec = load_district(str(schema))
    for k, ev in enumerate(spec.electric_vehicles):
        ev.battery.initial_soc = 0.1 + 0.1 * k
    tab = spec.episode_tables(0)
    E = 4
    rng = np.random.RandomState(seed)
    drift = rng.normal(1.0, 0.2, size=(tab.n_steps, len(spec.electric_vehicles))).astype(np.float32)
    eng = StepEngine(tab, E, reward='Electric_Vehicles_Reward_Function', detail=True, ev_drift=drift, charger_detail=True)
    o = FlexDistrictOracle(spec, tab, E, reward='Electric_Vehicles_Reward_Function', drift=drift.astype(np.float64))
    o.reset()
    lo, hi = spec.action_limits()
    flips = 0
    for t in range(150):
        a = rng.uniform(lo[:, None], hi[:, None], size=(len(lo), E)).astype(np.float32)
        a[rng.uniform(size=a.shape) < 0.15] = 0.0
        eng.step(torch.from_numpy(a).cuda())
        out = o.step(a)
        np.testing.assert_allclose(eng.ev_state[0].cpu().numpy(), out['ev_soc'], rtol=3e-4, atol=3e-4, err_msg=f'ev_soc t={t}')
        np.testing.assert_allclose(eng.charger_out[0].cpu().numpy(), out['charger_consumption'], rtol=3e-4, atol=3e-4, err_msg=f'charger t={t}')
        np.testing.assert_allclose(eng.net.cpu().numpy(), out['net'], rtol=3e-4, atol=3e-4, err_msg=f'net t={t}')
        rw = eng.reward_bldg.cpu().numpy()
        flips += int((np.abs(rw - out['reward']) > 1e-3 + 1e-3 * np.abs(out['reward'])).sum())
    assert flips <= 6, flips           # hard SoC thresholds of the reward: float32 vs float64 on a free-running trajectory


def test_observation_tensor_with_episode_offsets_on_the_ev_district():
    """Per-env-block windows + observation tensor: each block's reset() row carries the charger observations of an episode that
    starts on its own schedule row (arrival SoC of a first connection, not the mid-connection 0), later rows the step variant;
    streaming KPIs stay finite."""
    from citylearn_amd.vector_env import VectorCityLearnEnv
    g = golden('g2022_evs')
    E = 512
    env = VectorCityLearnEnv(g.schema_path, E, episode_time_steps=48, env_episode_offsets=[0, 96], observations='tensor',
                             observation_mode='reference', kpi=True)
    obs, _ = env.reset()
    names = [n for l in env.observation_names for n in l]
    ft = env.tables.flex
    soc_cols = [i for i, n in enumerate(names) if n.startswith('connected_electric_vehicle_at_charger_') and n.endswith('_soc')]
    assert len(soc_cols) == 8
    for blk, row in enumerate((0, 96)):
        expect = np.array([ft.reset_observations[names[c]][row] for c in soc_cols])
        got = obs[blk * 256, soc_cols].cpu().numpy()
        np.testing.assert_allclose(got, expect, rtol=1e-6, atol=1e-6)
        assert torch.equal(obs[blk * 256], obs[blk * 256 + 255])
    a = torch.full((env.n_act_cols, E), 0.25, device='cuda')
    for t in range(1, 6):
        obs, r, *_ = env.step(a)
        for blk, row in enumerate((0, 96)):
            expect = np.array([ft.observations[names[c]][row + t] for c in soc_cols])
            np.testing.assert_allclose(obs[blk * 256, soc_cols].cpu().numpy(), expect, rtol=1e-6, atol=1e-6)
    while not env.terminated:
        env.step(a)
    building, district = env.evaluate()
    for k in ('electricity_consumption_total', 'cost_total', 'carbon_emissions_total', 'ramping_average', 'daily_peak_average'):
        assert k in district and torch.isfinite(district[k]).all() and (district[k] > 0).all(), k


@pytest.mark.parametrize('reward', ['MARL', 'Electric_Vehicles_Reward_Function'])
def test_flex_rollout_equals_single_steps(reward):
    """`StepEngine.rollout` on a district with flexible loads (`cl_rollout_seq_f32`): the K-step launch sequence leaves the
    same building / EV / washing-machine state, last-step outputs and episode return as K `step()` calls, for open-loop
    actions and for the on-device Philox policy (host-side restatement of the same stream)."""
    from citylearn_amd.engine import StepEngine
    from citylearn_amd import _lib
    g = golden('g2022_evs')
    spec = g.spec()
    tab = spec.episode_tables(0)
    E, K, seed = 64, 36, 11
    low, high = spec.action_limits()
    lib = _lib.load()
    u = np.array([[[lib.cl_philox_uniform(seed, e, c, t) for e in range(E)] for c in range(len(low))] for t in range(K)], dtype=np.float32)
    acts = torch.from_numpy((low[None, :, None] + u * (high - low)[None, :, None]).astype(np.float32)).cuda()
    kw = dict(reward=reward, ev_seed=3)
    a, b, c = StepEngine(tab, E, **kw), StepEngine(tab, E, **kw), StepEngine(tab, E, **kw)
    ret_a = torch.zeros(E, device='cuda')
    for k in range(K):
        a.step(acts[k])
        ret_a += a.district_reward
    ret_b, ret_c = torch.zeros(E, device='cuda'), torch.zeros(E, device='cuda')
    b.rollout(K, actions=acts, ret_env=ret_b)
    c.set_action_limits(low, high)
    c.rollout(K, seed=seed, ret_env=ret_c)
    torch.cuda.synchronize()
    assert a.t == b.t == c.t == K
    assert float(a.ev_state[0].std(dim=1).max()) > 0.05                 # per-env actions did move the EVs apart
    for other, tol in ((b, 0.0), (c, 2e-5)):                       # the device policy rounds a = fma(u, span, low) once, the host twice
        for x, y in ((other.state, a.state), (other.ev_state, a.ev_state), (other.wm_state, a.wm_state),
                     (other.out_bldg[:2], a.out_bldg[:2]), (other.out_env, a.out_env)):
            torch.testing.assert_close(x, y, rtol=tol, atol=tol * 10)
    torch.testing.assert_close(ret_b, ret_a, rtol=1e-6, atol=1e-4)
    torch.testing.assert_close(ret_c, ret_a, rtol=1e-4, atol=1e-2)
    d = StepEngine(tab, E, **kw)
    d.set_action_limits(low, high)
    d.rollout(K, seed=seed + 1)
    assert not torch.equal(d.ev_state, c.ev_state)


def test_flex_rollout_replays_from_a_hip_graph():
    """`cl_rollout_seq_f32` only enqueues kernels on the caller's stream, so a whole K-step rollout of an EV district can be
    captured once and replayed: same state as the eager call."""
    from citylearn_amd.engine import StepEngine
    g = golden('g2022_evs')
    spec = g.spec()
    tab = spec.episode_tables(0)
    E, K = 256, 24
    low, high = spec.action_limits()
    eng, ref = StepEngine(tab, E, reward='MARL', ev_seed=3), StepEngine(tab, E, reward='MARL', ev_seed=3)
    for e in (eng, ref):
        e.set_action_limits(low, high)
    ret, ret_ref = torch.zeros(E, device='cuda'), torch.zeros(E, device='cuda')
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        eng.rollout(2, seed=1)                      # allocates the policy plane outside the capture
        eng.reset()
        stream.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=stream):
            eng.rollout(K, seed=9, ret_env=ret, t0=0)
        for _ in range(2):                          # replaying twice from a fresh episode gives the same episode twice
            eng.reset()
            ret.zero_()
            graph.replay()
        stream.synchronize()
    ref.rollout(K, seed=9, ret_env=ret_ref)
    torch.cuda.synchronize()
    assert torch.equal(eng.state, ref.state) and torch.equal(eng.ev_state, ref.ev_state) and torch.equal(eng.wm_state, ref.wm_state)
    assert torch.equal(ret, ret_ref) and float(ret.abs().sum()) > 0
