"""Observation epilogue `cl_observe_f32` on the GPU: against the host statement of the same affine map
(`ObservationTables.host_row`), and -- through `CityLearnEnv` / `NormalizedObservationWrapper` /
`VectorCityLearnEnv(observations='tensor')` -- against the observations the reference itself returned
(tests/golden/*/observations.npz).  GPU only."""
import numpy as np
import pytest
import torch

from golden_util import golden

pytestmark = pytest.mark.gpu


def _engine(name, n_env, detail=True):
    from citylearn_amd.engine import StepEngine
    g = golden(name)
    spec = g.spec()
    tables = spec.episode_tables(0)
    return g, spec, tables, StepEngine(tables, n_env, reward='RewardFunction', detail=detail)


@pytest.mark.parametrize('n_env,n_cols,n_dep', [(64, 52, 9), (100, 476, 34), (256, 527, 80), (68, 1500, 90), (192, 2600, 300), (4, 3, 3),
                                                 (100, 34, 34), (68, 64, 64), (260, 18, 18)])       # all columns dependent: the plane-transpose kernel
def test_kernel_matches_host_statement(n_env, n_cols, n_dep):
    """Synthetic column maps: single-segment (16-byte store path) and multi-segment shapes, ragged env tile, more
    dependent columns in a segment than the LDS staging holds (direct-read fallback)."""
    from citylearn_amd import abi
    from citylearn_amd.observations import ObservationTables, SRC_OUT, SRC_STATE, SRC_TEMP
    from citylearn_amd.observe import ObservationWriter
    from citylearn_amd.dynamics import LSTMStage
    g, spec, tables, eng = _engine('g2023_p2', n_env)
    stage = LSTMStage(spec, tables, eng)
    rng = np.random.RandomState(n_cols)
    B = eng.n_bldg
    eng.state.copy_(torch.from_numpy(rng.uniform(-1, 1, eng.state.shape).astype('float32')))
    eng.out_bldg.copy_(torch.from_numpy(rng.uniform(-5, 5, eng.out_bldg.shape).astype('float32')))
    stage.indoor_temp.copy_(torch.from_numpy(rng.uniform(15, 30, stage.indoor_temp.shape).astype('float32')))
    table = rng.uniform(-2, 2, (5, n_cols))
    src = np.full(n_cols, -1, dtype=np.int32)
    scale = np.zeros(n_cols, dtype=np.float32)
    for c in rng.choice(n_cols, size=n_dep, replace=False):
        kind = rng.randint(3)
        plane = rng.randint(abi.CL_NS) if kind == SRC_STATE else rng.randint(abi.CL_NO) if kind == SRC_OUT else 0
        src[c] = (kind << 28) | (plane << 20) | rng.randint(B)
        scale[c] = rng.uniform(0.1, 3.0)
    ot = ObservationTables(table, src, scale, needs_detail=False)
    w = ObservationWriter(eng, ot, stage)
    st, ob, tp = eng.state.cpu().numpy(), eng.out_bldg.cpu().numpy(), stage.indoor_temp.cpu().numpy()
    t32 = table.astype('float32').astype('float64')
    dense = torch.empty((n_env, n_cols), dtype=torch.float32, device='cuda')     # unpadded rows: scalar-store path
    for row in (0, 3):
        got = w.write(row).cpu().numpy()
        assert got.shape == (n_env, n_cols)
        assert np.array_equal(got, w.write(row, out=dense).cpu().numpy())
        assert not w._buffer[:, n_cols:].any()
        wide = torch.full((n_env, n_cols + 9), 7.0, dtype=torch.float32, device='cuda')     # view into a wider buffer
        assert np.array_equal(got, w.write(row, out=wide[:, :n_cols]).cpu().numpy())
        assert bool((wide[:, (n_cols + 3) // 4 * 4:] == 7.0).all())                        # nothing beyond the pad is touched
        deps, nd = w._deps, w.n_deps
        w._deps, w.n_deps = None, -1                                                       # no host list: device column map
        assert np.array_equal(got, w.write(row).cpu().numpy())
        w._deps, w.n_deps = deps, nd
        eng.tuning.obs_variant, eng.tuning.obs_rows = 1, 0                                                  # row-wise kernel on every shape
        assert np.array_equal(got, w.write(row).cpu().numpy())
        assert np.array_equal(got, w.write(row, out=dense).cpu().numpy())
        for rows in (0, 4, 8, 32, 64):                                                     # LDS-tile kernel, block heights
            eng.tuning.obs_variant, eng.tuning.obs_rows = 2, rows
            assert np.array_equal(got, w.write(row).cpu().numpy())
        eng.tuning.obs_variant, eng.tuning.obs_rows = 3, 0                                                 # wave-independent kernel
        assert np.array_equal(got, w.write(row).cpu().numpy())
        eng.tuning.obs_variant, eng.tuning.obs_rows = 0, 0
        for e in (0, 1, n_env // 2, n_env - 1):
            want = ObservationTables(t32, src, scale, False).host_row(row, st[:, :, e], ob[:, :, e], tp[:, e])
            np.testing.assert_allclose(got[e], want, rtol=1e-6, atol=1e-6)
    # every env row of an all-exogenous write is the table row, bit for bit
    assert np.array_equal(w.write(0).cpu().numpy(), np.broadcast_to(table[0].astype('float32'), (n_env, n_cols)))


def test_observe_validates_arguments():
    import ctypes
    from citylearn_amd import _lib, abi
    from citylearn_amd.observations import ObservationLayout
    from citylearn_amd.observe import ObservationWriter
    g, spec, tables, eng = _engine('g2022_all', 64)
    ot = ObservationLayout(spec, 'current').episode(tables)
    w = ObservationWriter(eng, ot)
    with pytest.raises(_lib.EngineError, match='row'):
        w.write(w.n_rows)
    rc = w.lib.cl_observe_f32(ctypes.byref(eng.dims), w.table.data_ptr(), None, None, None, -1, None, None, None, None, 0, w.obs.data_ptr(),
                              w.n_cols, w.pitch, w.n_rows, 1, 0, None)
    assert rc == abi.CL_ENULL
    rc = w.lib.cl_observe_f32(ctypes.byref(eng.dims), w.table.data_ptr() + 4, None, None, None, -1, None, None, None, None, 0, w.obs.data_ptr(),
                              w.n_cols, w.pitch, w.n_rows, 0, abi.CLOB_ALL_EXOGENOUS, None)
    assert rc == abi.CL_EALIGN
    rc = w.lib.cl_observe_f32(ctypes.byref(eng.dims), w.table.data_ptr(), None, None, None, -1, None, None, None, None, 0, w.obs.data_ptr(),
                              w.n_cols, w.n_cols - 1, w.n_rows, 0, abi.CLOB_ALL_EXOGENOUS, None)
    assert rc == abi.CL_EINVAL
    from citylearn_amd.observe import ObsDep
    bad = (ObsDep * 1)(ObsDep(w.n_cols, 0, 1.0))                                         # column outside the table
    rc = w.lib.cl_observe_f32(ctypes.byref(eng.dims), w.table.data_ptr(), w.col_src.data_ptr(), w.col_scale.data_ptr(),
                              ctypes.cast(bad, ctypes.c_void_p), 1, eng.state.data_ptr(), eng.out_bldg.data_ptr(), None, None, 0,
                              w.obs.data_ptr(), w.n_cols, w.pitch, w.n_rows, 1, 0, None)
    assert rc == abi.CL_EINVAL
    from citylearn_amd.engine import StepEngine
    lean = StepEngine(tables, 64, reward='RewardFunction', detail=False)
    spec2 = g.spec()
    for b in spec2.buildings:
        b.observation_metadata['electrical_storage_electricity_consumption'] = True
    with pytest.raises(ValueError, match='detail'):
        ObservationWriter(lean, ObservationLayout(spec2, 'current').episode(tables))


def _actions(g, env, t):
    a = [float(x) for x in g.ref['actions'][t]]
    if env.central_agent:
        return [a]
    out, p = [], 0
    for names in env.action_names:
        out.append(a[p:p + len(names)]); p += len(names)
    return out


@pytest.mark.parametrize('name', ['g2022_all', 'g2020_cz1', 'g2023_p2', 'g2023_heat', 'g2020_15min', 's_2021', 's_2023_p3'])
def test_env_returns_the_reference_observations(name):
    """reset()/step() observations, observation_space and the NormalizedObservationWrapper view of `CityLearnEnv`
    equal what the reference returned for the same schema and actions (reference semantics)."""
    from citylearn_amd.citylearn import CityLearnEnv
    from citylearn_amd.wrappers import NormalizedObservationWrapper
    g = golden(name)
    o = g.obs
    env = CityLearnEnv(g.schema_path)
    wrapped = NormalizedObservationWrapper(env)
    assert env.observation_names == g.obs_facts['observation_names']
    assert wrapped.observation_names == g.obs_facts['norm_observation_names']
    np.testing.assert_array_equal(np.concatenate([s.low for s in env.observation_space]), o['space_low'].astype('float32'))
    np.testing.assert_array_equal(np.concatenate([s.high for s in env.observation_space]), o['space_high'].astype('float32'))
    np.testing.assert_array_equal(np.concatenate([s.high for s in wrapped.observation_space]), o['norm_space_high'].astype('float32'))
    flat = lambda ll: np.array([x for l in ll for x in l])
    obs, _ = wrapped.reset()
    np.testing.assert_allclose(flat(obs), o['obs_norm'][0], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(flat(env.observations), o['obs'][0], rtol=1e-6, atol=1e-6)
    for t in range(40):
        obs, *_ = wrapped.step(_actions(g, env, t))
        np.testing.assert_allclose(flat(obs), o['obs_norm'][t + 1], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(flat(env.observations), o['obs'][t + 1], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize('name', ['g2022_all', 'g2020_cz1', 'g2023_p2', 'g2020_15min', 's_2023_p3'])
@pytest.mark.parametrize('normalize', [False, True])
def test_vector_env_observation_tensor(name, normalize):
    """`VectorCityLearnEnv(observations='tensor')`: exogenous columns equal the reference's returned observations, the
    env-dependent ones equal the reference's own series of the step just simulated (free-running fp32: 1e-3)."""
    from citylearn_amd.vector_env import VectorCityLearnEnv
    g = golden(name)
    o = g.obs
    kw = {'reward_function': 'citylearn.reward_function.RewardFunction'} if g.facts['reward_type'] == 'ComfortReward' else {}
    env = VectorCityLearnEnv(g.schema_path, 64, observations='tensor', normalize_observations=normalize, **kw)
    lay = env.layout
    lo, hi = lay.limits()
    ref = o['obs_norm' if normalize else 'obs']
    assert env.observation_names == g.obs_facts['norm_observation_names' if normalize else 'observation_names']
    obs, _ = env.reset()
    assert tuple(obs.shape) == (64, lay.n_cols)
    np.testing.assert_allclose(obs[5].cpu().numpy(), ref[0], rtol=1e-5, atol=1e-5)
    acts = torch.from_numpy(g.ref['actions']).cuda()
    dep = env.writer.col_src.cpu().numpy() >= 0
    for t in range(60):
        obs, *_ = env.step(acts[t][:, None].expand(-1, 64).contiguous())
        got = obs.cpu().numpy()
        assert np.array_equal(got[0], got[63])                     # identical actions -> identical envs
        np.testing.assert_allclose(got[0][~dep], ref[t + 1][~dep], rtol=1e-5, atol=1e-5)
        for c in np.nonzero(dep)[0]:
            i, k = lay.columns[c]
            if k.endswith('_delta'):
                sp = env.district_spec.buildings[i].series[k.replace('_delta', '_set_point')][t]
                want = float(o['robs_indoor_dry_bulb_temperature'][t, i]) - float(sp)
            else:
                want = float(o[f'robs_{k}'][t, i])
            if normalize:
                want = (want - lo[c]) / (hi[c] - lo[c])
            assert got[0][c] == pytest.approx(want, rel=1e-3, abs=1e-3), (t, i, k)


@pytest.mark.parametrize('name,normalize', [('g2022_all', False), ('g2022_all', True), ('g2023_p2', True), ('g2020_cz1', False)])
def test_compact_observations_expand_to_the_observation_tensor(name, normalize):
    """`VectorCityLearnEnv(observations='compact')` -- one shared row + the `[n_envs, n_dep]` matrix of the env-dependent columns --
    expands (`materialize`) to exactly what `observations='tensor'` writes, at reset and after every step, while moving ~7 % of
    the bytes (building.py:1115-1219: every other column is the same for all envs of the batch)."""
    from citylearn_amd.vector_env import VectorCityLearnEnv
    g = golden(name)
    E = 132
    full = VectorCityLearnEnv(g.schema_path, E, observations='tensor', normalize_observations=normalize)
    comp = VectorCityLearnEnv(g.schema_path, E, observations='compact', normalize_observations=normalize)
    o_full, _ = full.reset()
    o_comp, _ = comp.reset()
    assert set(o_comp) == {'shared', 'dependent', 'columns'} and o_comp['dependent'].shape == (E, len(o_comp['columns']))
    assert 0 < o_comp['dependent'].shape[1] < 0.4 * o_full.shape[1]      # 34 of 476 columns (2022), 18 of 54 under the 2023 central agent
    assert torch.equal(comp.materialize(o_comp), o_full)
    gen = torch.Generator(device='cuda').manual_seed(4)
    for t in range(20):
        a = full.sample_actions(gen)
        o_full = full.step(a)[0]
        o_comp = comp.step(a)[0]
        assert torch.equal(comp.materialize(o_comp), o_full), t


@pytest.mark.parametrize('E,tuning,offsets', [(65536, None, False), (772, dict(vec=4, lean_variant=2), False), (516, None, False),
                                               (1024, dict(vec=4, lean_variant=2), True)])
@pytest.mark.parametrize('kind', ['RewardFunction', 'MARL'])
@pytest.mark.parametrize('normalize', [False, True])
def test_step_observe_matches_step_then_observe(E, tuning, offsets, kind, normalize):
    """`cl_step_observe_f32` (`StepEngine.step_observe`): the compact observation of row t + 1 written by the step launch itself where
    the step is one lean launch at four envs per lane (65 536 envs; forced at 772 -- ragged last tile -- and with per-env-block episode
    offsets), by the two launches otherwise (516 envs; MARL, whose reward plane is finished after the tile is filled).  State, outputs
    and the observation matrix are bit-identical to `step` followed by `ObservationWriter.write` either way."""
    from citylearn_amd.engine import StepEngine
    from citylearn_amd.observations import ObservationLayout
    from citylearn_amd.observe import ObservationWriter
    g = golden('g2022_all')
    spec = g.spec()
    if offsets:
        tab = spec.episode_tables(0, window=(0, 400))
        row0 = np.array([0, 37, 5, 111], dtype=np.int32)
    else:
        tab, row0 = spec.episode_tables(0), None
    ot = ObservationLayout(spec, 'current', normalize).episode(tab, reset_table=offsets)
    dep_tables, _ = ot.compact()
    kw = dict(reward=kind, tuning=tuning)
    if offsets:
        kw.update(env_row0=row0, n_steps=200)
    a_eng, b_eng = StepEngine(tab, E, **kw), StepEngine(tab, E, **kw)
    wa, wb = ObservationWriter(a_eng, dep_tables, None), ObservationWriter(b_eng, dep_tables, None)
    assert wa.n_deps == wa.n_cols > 0
    gen = torch.Generator(device='cuda').manual_seed(E)
    for t in range(12):
        act = torch.rand((a_eng.n_act_cols, E), device='cuda', generator=gen) * 2 - 1
        a_eng.step(act, t)
        ref = wa.write(t + 1).clone()
        got = b_eng.step_observe(act, wb, t)
        assert torch.equal(a_eng.state, b_eng.state) and torch.equal(a_eng.out_bldg[:2], b_eng.out_bldg[:2]), t
        assert torch.equal(a_eng.out_env, b_eng.out_env), t
        assert torch.equal(got, ref), t
        assert torch.equal(wb._buffer[:, wb.n_cols:], torch.zeros_like(wb._buffer[:, wb.n_cols:]))        # pad columns stay zero
    assert got.abs().sum().item() > 0


@pytest.mark.parametrize('E,tuning', [(65536, None), (772, dict(vec=2)), (516, None)])
def test_step_observe_under_the_f64_chain(E, tuning):
    """`CLD_F64_CHAIN` through `cl_step_observe_f32`: the lean chain launch fills the observation tile itself
    (`cl_step_lean_obs_chain_kernel`) -- bit-identical to the chain `step` followed by `ObservationWriter.write`."""
    from citylearn_amd.engine import StepEngine
    from citylearn_amd.observations import ObservationLayout
    from citylearn_amd.observe import ObservationWriter
    spec = golden('g2022_all').spec()
    tab = spec.episode_tables(0)
    dep_tables, _ = ObservationLayout(spec, 'current', False).episode(tab).compact()
    a_eng, b_eng = StepEngine(tab, E, f64_maps='chain', tuning=tuning), StepEngine(tab, E, f64_maps='chain', tuning=tuning)
    wa, wb = ObservationWriter(a_eng, dep_tables, None), ObservationWriter(b_eng, dep_tables, None)
    b_eng.trace_kernels()
    gen = torch.Generator(device='cuda').manual_seed(E)
    for t in range(12):
        act = torch.rand((a_eng.n_act_cols, E), device='cuda', generator=gen) * 2 - 1
        a_eng.step(act, t)
        ref = wa.write(t + 1).clone()
        got = b_eng.step_observe(act, wb, t)
        assert torch.equal(a_eng.state, b_eng.state) and torch.equal(a_eng.out_bldg[:2], b_eng.out_bldg[:2]), t
        assert torch.equal(a_eng.out_env, b_eng.out_env), t
        assert torch.equal(got, ref), t
    assert b_eng.last_kernels.startswith('cl_step_lean_obs_chain_kernel<') and '+' not in b_eng.last_kernels, b_eng.last_kernels
    assert got.abs().sum().item() > 0


@pytest.mark.parametrize('f64', [None, False])
@pytest.mark.parametrize('name,E,tuning,kernel', [('g2020_cz1', 65536, None, 'cl_step_full_tp_obs_kernel<'), ('g2020_cz1', 4996, dict(full_variant=5), 'cl_step_full_tp_obs_kernel<'),
                                                  ('g2020_cz1', 772, None, 'cl_step_full_obs_kernel<'), ('s_2020_cz3', 516, None, 'cl_step_full_obs_kernel<'),
                                                  ('g2020_15min', 1284, dict(full_variant=5), 'cl_step_full_tp_obs_kernel<')])
def test_step_observe_on_thermal_districts_is_one_launch(name, E, tuning, kernel, f64):
    """`cl_step_observe_f32` on thermal districts (heat pump, heater, tanks: the 2020 schemas) -- the thermal step kernels of
    cl_full.h write the compact observation of row t + 1 themselves (battery AND tank states of charge, net: building.py:1115-1219, 1336-1481 return
    them from one `step`), the multi-tile kernel (9 x 65 536; forced at a ragged 4 996 / 1 284) and the one-tile kernel at one env per lane, under
    the default precision model and the all-fp32 map.  ONE launch, and state, outputs, district sums and the observation matrix bit-identical to
    `step` followed by `ObservationWriter.write`; pad columns zero."""
    from citylearn_amd.engine import StepEngine
    from citylearn_amd.observations import ObservationLayout
    from citylearn_amd.observe import ObservationWriter
    spec = golden(name).spec()
    tab = spec.episode_tables(0)
    dep_tables, _ = ObservationLayout(spec, 'current', False).episode(tab).compact()
    a_eng, b_eng = StepEngine(tab, E, f64_maps=f64, tuning=tuning), StepEngine(tab, E, f64_maps=f64, tuning=tuning)
    assert not a_eng.lean
    wa, wb = ObservationWriter(a_eng, dep_tables, None), ObservationWriter(b_eng, dep_tables, None)
    assert wa.n_deps == wa.n_cols > 17                       # more than the battery + net columns: tank states of charge too
    b_eng.trace_kernels()
    low, high = spec.action_limits()
    lo, hi = torch.from_numpy(low).cuda()[:, None], torch.from_numpy(high).cuda()[:, None]
    gen = torch.Generator(device='cuda').manual_seed(E)
    for t in range(10):
        act = (lo + torch.rand((a_eng.n_act_cols, E), device='cuda', generator=gen) * (hi - lo)).contiguous()
        a_eng.step(act, t)
        ref = wa.write(t + 1).clone()
        got = b_eng.step_observe(act, wb, t)
        assert torch.equal(a_eng.state, b_eng.state) and torch.equal(a_eng.out_bldg[:2], b_eng.out_bldg[:2]), t
        assert torch.equal(a_eng.out_env, b_eng.out_env), t
        assert torch.equal(got, ref), (t, (got - ref).abs().max().item())
        assert torch.equal(wb._buffer[:, wb.n_cols:], torch.zeros_like(wb._buffer[:, wb.n_cols:]))
    assert b_eng.last_kernels.startswith(kernel) and '+' not in b_eng.last_kernels, b_eng.last_kernels
    assert got.abs().sum().item() > 0
    # MARL finishes its reward plane after the sweep the tile is filled in: two launches, same result
    m_a, m_b = StepEngine(tab, E, reward='MARL', f64_maps=f64, tuning=tuning), StepEngine(tab, E, reward='MARL', f64_maps=f64, tuning=tuning)
    w_a, w_b = ObservationWriter(m_a, dep_tables, None), ObservationWriter(m_b, dep_tables, None)
    m_b.trace_kernels()
    act = (lo + torch.rand((m_a.n_act_cols, E), device='cuda', generator=gen) * (hi - lo)).contiguous()
    m_a.step(act, 0)
    assert torch.equal(m_b.step_observe(act, w_b, 0), w_a.write(1)) and '_obs_kernel' not in m_b.last_kernels, m_b.last_kernels


def test_degraded_capacity_is_not_an_observation_source_under_the_chain():
    """under CLD_F64_CHAIN the CLS_B_DEGCAP plane carries `capacity - degraded_capacity`; `cl_step_observe_f32` refuses a column fed by it."""
    import ctypes
    from citylearn_amd import _lib, abi
    from citylearn_amd.engine import StepEngine
    from citylearn_amd.observations import ObservationLayout
    from citylearn_amd.observe import ObservationWriter
    spec = golden('g2022_all').spec()
    tab = spec.episode_tables(0)
    dep_tables, _ = ObservationLayout(spec, 'current', False).episode(tab).compact()
    eng = StepEngine(tab, 64)
    assert eng.f64_chain
    w = ObservationWriter(eng, dep_tables, None)
    w._deps[0].src = (abi.CLOB_KIND_STATE << 28) | (abi.CLS_B_DEGCAP << 20) | 0
    with pytest.raises(_lib.EngineError) as e:
        eng.step_observe(torch.zeros((eng.n_act_cols, 64), device='cuda'), w, 0)
    assert e.value.code == abi.CL_EINVAL and 'CLS_B_DEGCAP' in str(e.value)
