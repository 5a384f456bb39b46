"""Fused K-step rollout (`cl_rollout_f32`, mode B) against K single steps (`cl_step_f32`) and the Philox policy
against its host-callable definition.  GPU only (the Philox host function is also checked on CPU in test_abi)."""
import ctypes

import numpy as np
import pytest
import torch

from golden_util import golden
from citylearn_amd import _lib, abi
from citylearn_amd.engine import StepEngine as _StepEngine

pytestmark = pytest.mark.gpu


def StepEngine(*args, f64_maps=False, **kw):
    """The fused-rollout instantiations and their tolerances against single steps were established on the all-fp32 battery map: engines of this
    module are built with `f64_maps=False` unless a test asks otherwise.  The default precision model (CLD_F64_CHAIN since round 6) in mode B is
    `test_fused_rollout_with_the_f64_chain` and `test_default_engine_rolls_out_under_the_chain`."""
    return _StepEngine(*args, f64_maps=f64_maps, **kw)


def _close(a, b, tol=2e-6):
    torch.testing.assert_close(a, b, rtol=tol, atol=tol)


@pytest.mark.parametrize('name,kind,E', [('g2022_all', 'RewardFunction', 512), ('g2022_all', 'MARL', 256),
                                         ('g2023_p2', 'SolarPenaltyReward', 192), ('g2020_cz1', 'IndependentSACReward', 128),
                                         ('g2022_all', 'MARL', 4), ('g2022_all', 'RewardFunction', 68), ('g2023_p2', 'RewardFunction', 36)])
def test_open_loop_rollout_equals_single_steps(name, kind, E):
    """Also batches that do not fill the last env tile (4, 68, 36 envs): the dead lanes of the ragged tile must not read the
    open-loop action tensor (it ends with the last live env)."""
    g = golden(name)
    spec = g.spec()
    tab = spec.episode_tables(0)
    K = 24
    low, high = spec.action_limits()
    gen = torch.Generator(device='cuda').manual_seed(9)
    lo, hi = torch.from_numpy(low).cuda(), torch.from_numpy(high).cuda()
    acts = lo[None, :, None] + torch.rand((K, len(low), E), device='cuda', generator=gen) * (hi - lo)[None, :, None]
    a, b = StepEngine(tab, E, reward=kind), StepEngine(tab, E, reward=kind)
    ret_ref = torch.zeros(E, device='cuda')
    for k in range(K):
        a.step(acts[k])
        ret_ref += a.district_reward
    ret = torch.zeros(E, device='cuda')
    b.rollout(K, actions=acts, ret_env=ret)
    _close(b.state, a.state)
    _close(b.out_bldg[:2], a.out_bldg[:2], 2e-5)
    _close(b.out_env, a.out_env, 1e-4)
    torch.testing.assert_close(ret, ret_ref, rtol=1e-5, atol=1e-3)
    assert b.t == a.t == K
    # continue from the rolled-out state: a second rollout of 12 steps from t = 24
    more = acts[:12].contiguous()
    for k in range(12):
        a.step(more[k])
    b.rollout(12, actions=more)
    _close(b.state, a.state)


def test_on_device_philox_policy_matches_host_definition():
    g = golden('g2022_all')
    spec = g.spec()
    tab = spec.episode_tables(0)
    E, K, seed = 128, 6, 5
    lib = _lib.load()
    low, high = spec.action_limits()
    u = np.array([[[lib.cl_philox_uniform(seed, e, c, t) for e in range(E)] for c in range(len(low))] for t in range(K)], dtype=np.float32)
    assert u.min() >= 0.0 and u.max() < 1.0 and abs(u.mean() - 0.5) < 0.02 and len(np.unique(u)) > 0.99 * u.size
    host_actions = torch.from_numpy((low[None, :, None] + u * (high - low)[None, :, None]).astype(np.float32)).cuda()
    a, b = StepEngine(tab, E), StepEngine(tab, E)
    b.set_action_limits(low, high)
    ret_a, ret_b = torch.zeros(E, device='cuda'), torch.zeros(E, device='cuda')
    a.rollout(K, actions=host_actions, ret_env=ret_a)
    b.rollout(K, seed=seed, ret_env=ret_b)
    _close(b.state, a.state)
    torch.testing.assert_close(ret_b, ret_a, rtol=1e-5, atol=1e-4)
    c = StepEngine(tab, E)
    c.set_action_limits(low, high)
    c.rollout(K, seed=seed + 1)
    assert not torch.equal(c.state, b.state)                      # a different seed is a different policy draw


def test_rollout_errors():
    g = golden('g2022_all')
    tab = g.spec().episode_tables(0)
    eng = StepEngine(tab, 64)
    with pytest.raises(ValueError):
        eng.rollout(4)                                            # on-device policy without action limits
    with pytest.raises(_lib.EngineError) as e:
        eng.rollout(10 ** 6, actions=None, seed=1) if eng.set_action_limits(*g.spec().action_limits()) is None else None
    assert e.value.code == abi.CL_ERANGE


@pytest.mark.parametrize('name', ['g2022_all', 'g2022_evs'])
def test_vector_env_rollout_matches_stepping(name):
    """`VectorCityLearnEnv.rollout` (fused kernel, or the launch sequence on the EV district) against `step()` fed with the host
    restatement of the same Philox policy stream: same episode return, same state, and the env carries on from there."""
    from citylearn_amd.vector_env import VectorCityLearnEnv
    g = golden(name)
    E, K, seed = 64, 12, 21
    kw = dict(reward_function='citylearn.reward_function.RewardFunction', ev_seed=5) if name == 'g2022_evs' else {}
    a, b = VectorCityLearnEnv(g.schema_path, E, **kw), VectorCityLearnEnv(g.schema_path, E, **kw)
    lib = _lib.load()
    low, high = a.action_low.cpu().numpy(), a.action_high.cpu().numpy()
    u = np.array([[[lib.cl_philox_uniform(seed, e, c, t) for e in range(E)] for c in range(len(low))] for t in range(K)], dtype=np.float32)
    acts = torch.from_numpy((low[None, :, None] + u * (high - low)[None, :, None]).astype(np.float32)).cuda()
    ret_ref = torch.zeros(E, device='cuda')
    for k in range(K):
        _, reward, *_ = a.step(acts[k])
        ret_ref += reward.sum(dim=0)
    ret = b.rollout(K, seed=seed)
    assert b.time_step == a.time_step == K
    torch.testing.assert_close(ret, ret_ref, rtol=1e-4, atol=1e-2)
    torch.testing.assert_close(b.engine.state, a.engine.state, rtol=2e-5, atol=2e-5)
    nxt = a.sample_actions(torch.Generator(device='cuda').manual_seed(1))
    ra, rb = a.step(nxt)[1], b.step(nxt)[1]
    torch.testing.assert_close(rb, ra, rtol=1e-4, atol=1e-3)
    with pytest.raises(RuntimeError, match='past the episode end'):
        b.rollout(b.time_steps)


@pytest.mark.parametrize('name,kind,B', [('g2022_all', 'MARL', 0), ('g2020_cz1', 'RewardFunction', 0), ('g2022_all', 'RewardFunction', 100),
                                         ('g2020_cz1', 'SolarPenaltyReward', 48)])
def test_rollout_with_streaming_kpis_and_large_districts(name, kind, B):
    """What the fused kernel does not hold in registers runs as the launch sequence `cl_rollout_seq_f32`: streaming KPI accumulators
    (CLD_KPI) and districts beyond 32 / 16 buildings (building-chunked launches).  K steps of it equal K calls of `cl_step_f32`
    bit for bit -- state, last outputs, episode return and every KPI accumulator."""
    from citylearn_amd.synthetic import tile_district
    spec = golden(name).spec()
    if B:
        spec = tile_district(spec, B)
    tab = spec.episode_tables(0)
    E, K, seed = 192, 30, 9
    kpi = B == 0
    a, b = StepEngine(tab, E, reward=kind, kpi=kpi), StepEngine(tab, E, reward=kind, kpi=kpi)
    low, high = spec.action_limits()
    cols = len(low)
    ret = torch.zeros(E, device='cuda')
    gen = torch.Generator(device='cuda').manual_seed(seed)
    lo, hi = torch.from_numpy(low).cuda(), torch.from_numpy(high).cuda()
    acts = lo[None, :, None] + torch.rand((K, cols, E), device='cuda', generator=gen) * (hi - lo)[None, :, None]
    b.rollout(K, actions=acts, ret_env=ret, fused=False if B else None)      # (large districts: the launch sequence asked for explicitly --
    ret_ref = torch.zeros(E, device='cuda')                                  #  by default they run the chunked fused kernel, tested below)
    for k in range(K):
        a.step(acts[k])
        ret_ref += a.district_reward
    assert torch.equal(b.state, a.state) and torch.equal(b.out_bldg, a.out_bldg) and torch.equal(b.out_env, a.out_env)
    torch.testing.assert_close(ret, ret_ref, rtol=1e-6, atol=1e-4)
    if kpi:
        assert torch.equal(b.kpi_bldg, a.kpi_bldg) and torch.equal(b.kpi_env, a.kpi_env)
        assert float(b.kpi_bldg.abs().sum()) > 0
    assert b.t == a.t == K
    if not B:
        # the on-device Philox policy through the same entry point: the host replays the stream (a = fma(u, high - low, low), here in
        # float64 and rounded once) and steps; one-ulp action differences are possible, hence tolerances
        lib = _lib.load()
        c, d = StepEngine(tab, E, reward=kind, kpi=True), StepEngine(tab, E, reward=kind, kpi=True)
        u = np.array([[[lib.cl_philox_uniform(seed, e, col, t) for e in range(E)] for col in range(cols)] for t in range(8)], dtype=np.float64)
        host = torch.from_numpy((low.astype(np.float64)[None, :, None] + u * (high - low).astype(np.float64)[None, :, None]).astype(np.float32)).cuda()
        d.set_action_limits(low, high)
        d.rollout(8, seed=seed)
        for k in range(8):
            c.step(host[k])
        torch.testing.assert_close(d.state, c.state, rtol=2e-5, atol=2e-5)
        torch.testing.assert_close(d.kpi_bldg, c.kpi_bldg, rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize('name,kind,B,E,tuning', [('g2022_all', 'RewardFunction', 1024, 256, None), ('g2020_cz1', 'RewardFunction', 1024, 128, None),
                                                  ('g2022_all', 'SolarPenaltyReward', 100, 260, None), ('g2020_cz1', 'IndependentSACReward', 48, 68, None),
                                                  ('g2022_all', 'RewardFunction', 70, 4096, dict(vec=2)), ('g2022_all', 'IndependentSACReward', 33, 64, dict(b_chunk=24))])
def test_chunked_fused_rollout_equals_single_steps(name, kind, B, E, tuning):
    """Mode B for building-chunked districts (round 5; BASELINE config 4's 1024 buildings): `cl_rollout_f32` cuts the district into workgroup
    rows of 32 battery + PV / 16 thermal buildings, keeps every unit's state in registers for the K steps and leaves the last step's chunk
    partial sums + each chunk's share of the K-step return to one `cl_finish_kernel` launch.  Against K calls of `cl_step_f32`
    (citylearn.py:978-1056 K times, district sums citylearn.py:1888-1918) within the fused kernel's tolerance (the tolerances of
    test_open_loop_rollout_equals_single_steps; district sums scaled with the district size); ragged last chunks (100 = 3 x 32 + 4, 48 = 3 x 16,
    33 = 24 + 9), ragged last env tiles (260, 68 envs), both pack widths; a second rollout continues from the state the first one left."""
    from citylearn_amd.synthetic import tile_district
    spec = tile_district(golden(name).spec(), B)
    tab = spec.episode_tables(0)
    K = 24
    low, high = spec.action_limits()
    gen = torch.Generator(device='cuda').manual_seed(B + E)
    lo, hi = torch.from_numpy(low).cuda(), torch.from_numpy(high).cuda()
    acts = lo[None, :, None] + torch.rand((K, len(low), E), device='cuda', generator=gen) * (hi - lo)[None, :, None]
    a, b = StepEngine(tab, E, reward=kind), StepEngine(tab, E, reward=kind, tuning=tuning)
    b.trace_kernels()
    ret_ref = torch.zeros(E, device='cuda', dtype=torch.float64)
    for k in range(K):
        a.step(acts[k])
        ret_ref += a.district_reward.double()
    ret = torch.full((E,), 3.0, device='cuda')                       # (the return is ADDED to what the caller passes in)
    b.rollout(K, actions=acts, ret_env=ret)
    # (round 6: thermal districts run the pack-generic unit of cl_full.h inside the K-step loop, cl_rollout_full_kernel -- two envs per lane on the fp32 map)
    thermal = name == 'g2020_cz1'
    assert (b.last_kernels == 'cl_rollout_full_kernel<2, true, 0, false>+cl_finish_kernel') if thermal else \
        ('cl_rollout_kernel' in b.last_kernels and b.last_kernels.endswith(', true, 0>+cl_finish_kernel')), b.last_kernels
    _close(b.state, a.state)
    _close(b.out_bldg[:2], a.out_bldg[:2], 2e-5)
    # district sums over B buildings: the per-building tolerance times the district size
    torch.testing.assert_close(b.out_env, a.out_env, rtol=1e-5, atol=1e-6 * B)
    torch.testing.assert_close(ret.double() - 3.0, ret_ref, rtol=1e-5, atol=2e-5 * B)
    assert b.t == a.t == K and b._pending_t is None
    more = acts[:6].contiguous()
    for k in range(6):
        a.step(more[k])
    b.rollout(6, actions=more)
    _close(b.state, a.state)
    torch.testing.assert_close(b.out_env, a.out_env, rtol=1e-5, atol=1e-6 * B)


def test_chunked_fused_rollout_policy_and_fallbacks():
    """The on-device Philox policy through the chunked kernel (same stream as the launch sequence draws: env, column, step), MARL on a
    chunked district (falls back to the launch sequence: the reward couples the buildings inside a step), and the room check."""
    from citylearn_amd.synthetic import tile_district
    spec = tile_district(golden('g2022_all').spec(), 256)
    tab = spec.episode_tables(0)
    E, K, seed = 512, 12, 31
    low, high = spec.action_limits()
    f, q = StepEngine(tab, E), StepEngine(tab, E)
    for e in (f, q):
        e.set_action_limits(low, high)
        e.trace_kernels()
    rf, rq = torch.zeros(E, device='cuda'), torch.zeros(E, device='cuda')
    f.rollout(K, seed=seed, ret_env=rf)
    q.rollout(K, seed=seed, ret_env=rq, fused=False)
    assert 'cl_rollout_kernel' in f.last_kernels and 'cl_rollout_kernel' not in q.last_kernels
    _close(f.state, q.state)
    torch.testing.assert_close(f.out_env, q.out_env, rtol=1e-5, atol=3e-4)
    torch.testing.assert_close(rf, rq, rtol=1e-5, atol=5e-3)
    m = StepEngine(tab, E, reward='MARL')
    m.set_action_limits(low, high)
    m.trace_kernels()
    m.rollout(4, seed=seed)
    assert 'cl_rollout_kernel' not in m.last_kernels and 'cl_marl_reward_kernel' in m.last_kernels
    with pytest.raises(_lib.EngineError) as err:
        m.rollout(4, seed=seed, fused=True)
    assert err.value.code == abi.CL_EINVAL and 'MARL' in str(err.value)


@pytest.mark.parametrize('name,B,E', [('g2022_all', 0, 192), ('g2020_cz1', 0, 128), ('g2022_all', 100, 68), ('g2020_cz1', 40, 64), ('g2023_p2', 0, 64), ('g2022_all', 0, 32768)])
def test_fused_rollout_with_the_f64_chain(name, B, E):
    """CLD_F64_CHAIN in mode B (`cl_rollout_kernel<.., PREC = 2>`: battery + PV and thermal districts, one workgroup row and building-chunked):
    K fused steps against K single steps of the same precision model -- the soc chain is float64 in both, so the battery state agrees to
    the last bits a differently contracted fp32 epilogue can move -- and a fixture's own actions keep the rolled-out soc on the reference."""
    from citylearn_amd.synthetic import tile_district
    g = golden(name)
    spec = tile_district(g.spec(), B) if B else g.spec()
    tab = spec.episode_tables(0)
    K = 24
    low, high = spec.action_limits()
    gen = torch.Generator(device='cuda').manual_seed(E)
    lo, hi = torch.from_numpy(low).cuda(), torch.from_numpy(high).cuda()
    acts = lo[None, :, None] + torch.rand((K, len(low), E), device='cuda', generator=gen) * (hi - lo)[None, :, None]
    if not B:
        acts[:, :, 0] = torch.from_numpy(g.ref['actions'][:K]).cuda()                 # env 0 replays the fixture
    a, b = StepEngine(tab, E, f64_maps='chain'), StepEngine(tab, E, f64_maps='chain')
    b.trace_kernels()
    for k in range(K):
        a.step(acts[k])
    ret = torch.zeros(E, device='cuda')
    b.rollout(K, actions=acts, ret_env=ret)
    # (two envs per lane where the batch fills the chip in whole rounds -- 32 768 envs, the C5 shard -- else one)
    lean = name == 'g2022_all'
    assert (('cl_rollout_kernel<2, ' if E >= 32768 else 'cl_rollout_kernel<1, ') if lean else 'cl_rollout_full_kernel<1, ') in b.last_kernels and \
        (', 2>' if lean else ', 2, ') in b.last_kernels, b.last_kernels
    _close(b.state, a.state)
    _close(b.out_bldg[:2], a.out_bldg[:2], 2e-5)
    torch.testing.assert_close(b.out_env, a.out_env, rtol=1e-5, atol=1e-6 * max(B, 17))
    if not B:
        has = torch.tensor([bl.electrical_storage.present for bl in spec.buildings], device='cuda')
        soc, ref = b.soc[:, 0][has].cpu().numpy(), g.ref['soc'][K - 1][has.cpu().numpy()]
        assert float(np.max(np.abs(soc - ref) / (1e-4 + 1e-4 * np.abs(ref)))) < 0.1


def test_sharded_rollouts_reproduce_the_unsharded_policy_stream():
    """`cl_dims.env_offset`: two shards of a 512-env batch (envs [0, 256) and [256, 512)), rolled out with the SAME seed and their
    shard offsets, reproduce the unsharded rollout env for env -- what makes the multi-GPU decomposition (one process per GPU,
    contiguous env ranges, no collective) independent of the number of ranks."""
    from citylearn_amd.parallel import shard_envs
    g = golden('g2022_all')
    spec = g.spec()
    tab = spec.episode_tables(0)
    low, high = spec.action_limits()
    E, K, seed = 512, 16, 77
    whole = StepEngine(tab, E, reward='MARL')
    whole.set_action_limits(low, high)
    ret = torch.zeros(E, device='cuda')
    whole.rollout(K, seed=seed, ret_env=ret)
    for rank in range(2):
        lo, hi = shard_envs(E, rank, 2)
        part = StepEngine(tab, hi - lo, reward='MARL', env_offset=lo)
        part.set_action_limits(low, high)
        r = torch.zeros(hi - lo, device='cuda')
        part.rollout(K, seed=seed, ret_env=r)
        assert torch.equal(part.state, whole.state[:, :, lo:hi]) and torch.equal(part.out_bldg[:2], whole.out_bldg[:2, :, lo:hi])
        assert torch.equal(r, ret[lo:hi])
    same_seed_no_offset = StepEngine(tab, 256, reward='MARL')
    same_seed_no_offset.set_action_limits(low, high)
    same_seed_no_offset.rollout(K, seed=seed)
    assert not torch.equal(same_seed_no_offset.state, whole.state[:, :, 256:])           # without the offset shard 1 would repeat shard 0's draws


def test_vector_env_rollout_keeps_streaming_kpis():
    """`VectorCityLearnEnv(kpi=True).rollout` (launch sequence with the KPI passes after every step): `evaluate()` afterwards equals
    the one of an env stepped through the same open-loop actions."""
    from citylearn_amd.vector_env import VectorCityLearnEnv
    g = golden('g2022_all')
    E, K = 128, 48
    a, b = VectorCityLearnEnv(g.schema_path, E, kpi=True), VectorCityLearnEnv(g.schema_path, E, kpi=True)
    gen = torch.Generator(device='cuda').manual_seed(2)
    acts = torch.stack([a.sample_actions(gen) for _ in range(K)])
    ret_ref = torch.zeros(E, device='cuda')
    for k in range(K):
        ret_ref += a.step(acts[k])[1].sum(dim=0)
    ret = b.rollout(K, actions=acts)
    torch.testing.assert_close(ret, ret_ref, rtol=1e-6, atol=1e-4)
    (bld_a, dis_a), (bld_b, dis_b) = a.evaluate(), b.evaluate()
    assert set(bld_a) == set(bld_b) and set(dis_a) == set(dis_b) and len(dis_a) >= 5
    for k in bld_a:
        torch.testing.assert_close(bld_b[k], bld_a[k], rtol=0, atol=0, equal_nan=True)
    for k in dis_a:
        torch.testing.assert_close(dis_b[k], dis_a[k], rtol=0, atol=0, equal_nan=True)


def test_step_many_is_k_steps_from_one_call():
    """`StepEngine.step_many` (one C call enqueuing k step launches, `cl_rollout_seq_f32` with open-loop actions) == k calls of `step`."""
    g = golden('g2022_all')
    tab = g.spec().episode_tables(0)
    E, K = 260, 9
    a, b = StepEngine(tab, E, reward='MARL'), StepEngine(tab, E, reward='MARL')
    acts = torch.rand((K, a.n_act_cols, E), device='cuda', generator=torch.Generator(device='cuda').manual_seed(4)) * 2 - 1
    for k in range(K):
        a.step(acts[k], 3 + k)
    b.step_many(acts, 3)
    assert b.t == a.t == 3 + K
    assert torch.equal(a.state, b.state) and torch.equal(a.out_bldg[:2], b.out_bldg[:2]) and torch.equal(a.out_env, b.out_env)


def test_default_engine_rolls_out_under_the_chain():
    """An engine built with default arguments steps AND rolls out under CLD_F64_CHAIN (one precision model for both modes): K fused steps
    equal K single steps of the same engine type."""
    g = golden('g2022_all')
    spec = g.spec()
    tab = spec.episode_tables(0)
    E, K = 256, 24
    b = _StepEngine(tab, E)
    assert b.f64_chain
    b.trace_kernels()
    low, high = spec.action_limits()
    b.set_action_limits(low, high)
    b.rollout(K, seed=3)
    assert 'cl_rollout_kernel' in b.last_kernels and b.last_kernels.rstrip('>').endswith(', 2'), b.last_kernels
    # (the policy stream itself is pinned by test_on_device_philox_policy_matches_host_definition: replay it through the open-loop path)
    c = _StepEngine(tab, E)
    c.set_action_limits(low, high)
    c.rollout(K, seed=3, fused=False)                      # the launch sequence: cl_policy_kernel + K x cl_step_f32
    _close(b.state, c.state, 2e-6)
    _close(b.out_bldg[:2], c.out_bldg[:2], 2e-5)


@pytest.mark.parametrize('f64', [False, 'chain'])
@pytest.mark.parametrize('name,B,E,kind', [('g2020_cz1', 0, 516, 'RewardFunction'), ('g2020_cz1', 0, 128, 'MARL'), ('g2023_p2', 0, 260, 'SolarPenaltyReward'),
                                           ('g2020_cz1', 40, 132, 'IndependentSACReward'), ('s_2023_p3', 0, 64, 'RewardFunction')])
def test_packed_thermal_rollout_against_the_scalar_unit_and_single_steps(name, B, E, kind, f64):
    """`cl_rollout_full_kernel` -- mode B for thermal / outage districts around `clv::unit_step`, the pack-generic arithmetic of the
    thermal STEP kernels (two envs per lane on the fp32 map, one under the float64 chain) -- against the scalar-unit rollout it replaces
    (`cl_tuning.full_variant = 1` -> `cl_rollout_kernel<1, true, 1>`) and against K single steps: state, net, reward, district sums and K-step
    return within the fused kernels' tolerance; one workgroup row and building-chunked; MARL (one LDS exchange per step); ragged env tiles;
    outage rows of the 2023 fixtures (steps 380 .. 410); open-loop actions and the on-device Philox policy."""
    from citylearn_amd.synthetic import tile_district
    g = golden(name)
    spec = tile_district(g.spec(), B) if B else g.spec()
    tab = spec.episode_tables(0)
    K = 24
    t0 = 385 if (name == 'g2023_p2') else 0
    low, high = spec.action_limits()
    gen = torch.Generator(device='cuda').manual_seed(E + len(kind))
    lo, hi = torch.from_numpy(low).cuda(), torch.from_numpy(high).cuda()
    acts = lo[None, :, None] + torch.rand((K, len(low), E), device='cuda', generator=gen) * (hi - lo)[None, :, None]
    a = StepEngine(tab, E, reward=kind, f64_maps=f64)
    b = StepEngine(tab, E, reward=kind, f64_maps=f64)
    c = StepEngine(tab, E, reward=kind, f64_maps=f64, tuning=dict(full_variant=1))
    b.trace_kernels(); c.trace_kernels()
    if t0:                                     # mid-episode start: some charge in every storage
        for e in (a, b, c):
            e.state[abi.CLS_B_SOC] = 0.5; e.state[abi.CLS_DS_SOC] = 0.3
    ret_ref = torch.zeros(E, device='cuda', dtype=torch.float64)
    for k in range(K):
        a.step(acts[k], t0 + k)
        ret_ref += a.district_reward.double()
    rb, rc = torch.zeros(E, device='cuda'), torch.zeros(E, device='cuda')
    b.rollout(K, actions=acts, ret_env=rb, t0=t0)
    c.rollout(K, actions=acts, ret_env=rc, t0=t0)
    marl = kind == 'MARL'                                # (one LDS exchange per step: its own instantiation, one env per lane)
    assert b.last_kernels.startswith(f'cl_rollout_full_kernel<{1 if f64 or marl else 2}, ') and f", {'true' if marl else 'false'}>" in b.last_kernels and \
        c.last_kernels.startswith('cl_rollout_kernel<1, true, 1, '), (b.last_kernels, c.last_kernels)
    nb = max(B, len(spec.buildings))
    for x in (b, c):
        _close(x.state, a.state)
        _close(x.out_bldg[:2], a.out_bldg[:2], 2e-5)
        torch.testing.assert_close(x.out_env, a.out_env, rtol=1e-5, atol=2e-6 * nb)
    torch.testing.assert_close(rb.double(), ret_ref, rtol=1e-5, atol=2e-5 * nb * K)
    torch.testing.assert_close(rb, rc, rtol=1e-5, atol=2e-5 * nb * K)
    # the on-device policy draws the same stream in both kernels
    p, q = StepEngine(tab, E, reward=kind, f64_maps=f64), StepEngine(tab, E, reward=kind, f64_maps=f64, tuning=dict(full_variant=1))
    for e in (p, q):
        e.set_action_limits(low, high)
    p.rollout(12, seed=9, t0=t0); q.rollout(12, seed=9, t0=t0)
    _close(p.state, q.state)
    _close(p.out_bldg[:2], q.out_bldg[:2], 2e-5)


@pytest.mark.parametrize('f64', [False, 'chain'])
def test_packed_rollout_with_six_action_columns(f64):
    """The packed thermal rollout keeps the Philox blocks of a building's first four action columns in LDS at two envs per lane (six at one) and
    redraws a fifth / sixth column's word every step: a district whose buildings act on all six columns (2023 phase-2 buildings given the two missing
    tanks, a heating device and their actions) -- same streams as the scalar-unit kernel and as the policy's host definition."""
    import copy
    import dataclasses
    spec = copy.deepcopy(golden('g2023_p2').spec())
    for b in spec.buildings:
        b.action_metadata = dict(b.action_metadata, cooling_storage=True, heating_storage=True, heating_device=True)
        b.cooling_storage = dataclasses.replace(b.dhw_storage)
        b.heating_storage = dataclasses.replace(b.dhw_storage)
        b.heating_device = dataclasses.replace(b.cooling_device)
    assert all(len(b.active_actions) == 6 for b in spec.buildings)
    tab = spec.episode_tables(0)
    E, K = 260, 11                                      # (t0 = 2: the first block covers steps 2 .. 3 only)
    low, high = spec.action_limits()
    p, q = StepEngine(tab, E, f64_maps=f64), StepEngine(tab, E, f64_maps=f64, tuning=dict(full_variant=1))
    p.trace_kernels()
    rp, rq = torch.zeros(E, device='cuda'), torch.zeros(E, device='cuda')
    for e, ret in ((p, rp), (q, rq)):
        e.set_action_limits(low, high)
        e.rollout(K, seed=77, t0=2, ret_env=ret)
    assert p.last_kernels.startswith(f'cl_rollout_full_kernel<{1 if f64 else 2}, false, '), p.last_kernels
    _close(p.state, q.state)
    _close(p.out_bldg[:2], q.out_bldg[:2], 2e-5)
    torch.testing.assert_close(rp, rq, rtol=1e-5, atol=1e-3)
    # ... and as K single steps on the actions the host definition of the stream gives (a sample of envs: one call per draw)
    a = StepEngine(tab, E, f64_maps=f64)
    lib = _lib.load()
    lib.cl_philox_uniform.restype = ctypes.c_float
    lib.cl_philox_uniform.argtypes = [ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32]
    sample = list(range(0, E, 37))
    u = np.array([[[lib.cl_philox_uniform(77, e, c, 2 + k) for e in sample] for c in range(len(low))] for k in range(K)], dtype=np.float32)
    acts = torch.zeros((K, len(low), E), device='cuda')
    acts[:, :, sample] = torch.from_numpy((low[None, :, None] + u * (high - low)[None, :, None]).astype(np.float32)).cuda()
    for k in range(K):
        a.step(acts[k], 2 + k)
    _close(p.state[:, :, sample], a.state[:, :, sample])
    _close(p.out_bldg[:2][:, :, sample], a.out_bldg[:2][:, :, sample], 2e-5)


@pytest.mark.parametrize('B,E,vec', [(64, 16384, 2), (70, 1028, 1), (1024, 256, 1), (1024, 1024, 2)])
def test_chunked_fused_rollout_under_the_chain_at_both_pack_widths(B, E, vec):
    """The building-chunked battery + PV rollout around the float64 chain: one env per lane where the launch is small, TWO where its 128-env workgroups
    fill the chip (round 6: `cl_rollout_kernel<2, false, 2, true, true, 2>`, 1024 x 1024 127.6 -> 107.9 us per 24 steps) -- against K single steps
    under the same precision model, open-loop actions and the on-device policy (whose stream does not depend on the pack width)."""
    from citylearn_amd.synthetic import tile_district
    spec = tile_district(golden('g2022_all').spec(), B)
    tab = spec.episode_tables(0)
    K = 24
    low, high = spec.action_limits()
    gen = torch.Generator(device='cuda').manual_seed(B * 7 + E)
    lo, hi = torch.from_numpy(low).cuda(), torch.from_numpy(high).cuda()
    acts = lo[None, :, None] + torch.rand((K, len(low), E), device='cuda', generator=gen) * (hi - lo)[None, :, None]
    a, b = StepEngine(tab, E, f64_maps='chain'), StepEngine(tab, E, f64_maps='chain')
    b.trace_kernels()
    ret_ref = torch.zeros(E, device='cuda', dtype=torch.float64)
    for k in range(K):
        a.step(acts[k])
        ret_ref += a.district_reward.double()
    ret = torch.zeros(E, device='cuda')
    b.rollout(K, actions=acts, ret_env=ret)
    assert b.last_kernels == f'cl_rollout_kernel<{vec}, false, 2, true, true, 2>+cl_finish_kernel', b.last_kernels
    _close(b.state, a.state)
    _close(b.out_bldg[:2], a.out_bldg[:2], 2e-5)
    torch.testing.assert_close(b.out_env, a.out_env, rtol=1e-5, atol=1e-6 * B)
    torch.testing.assert_close(ret.double(), ret_ref, rtol=1e-5, atol=2e-5 * B)
    p, q = StepEngine(tab, E, f64_maps='chain'), StepEngine(tab, E, f64_maps='chain', tuning=dict(vec=3 - vec))
    for e in (p, q):
        e.set_action_limits(low, high)
        e.rollout(12, seed=5)
    _close(p.state, q.state)
    _close(p.out_bldg[:2], q.out_bldg[:2], 2e-5)
