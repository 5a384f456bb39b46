"""Parity at the sizes BASELINE.json's configs name, where the launch geometry (`vec`, `b_chunk`, `n_chunks`, grid size,
kernel template) takes values the small-batch parity tests never reach.  GPU only.

* C3 / thermal districts at 65 536 envs (general kernel, FULL): replication + checksum properties against a 512-env engine that
  the oracle / reference tests cover (tests/test_gpu_parity.py), on the 2023 outage schema (3 buildings) and the 2020 schema
  (9 buildings: heat pump, heater, tanks).
* C4 per-GPU shard, 1024 buildings x 1024 envs (building-chunked grid + cl_finish_kernel + cl_marl_reward_kernel) against the
  C oracle, teacher-forced, all four fused rewards, both device sets.
* C5 kernel template (`cl_rollout_kernel<2, false, 2>`, selected from 131 072 envs) against K single steps.
Reference lines: building.py:1500-1634 (apply_actions), citylearn.py:1888-1918 (district sums), reward_function.py:65-214.
"""
from functools import lru_cache

import numpy as np
import pytest
import torch

from golden_util import golden, check_worst, coupled_reward_tolerance
from citylearn_amd import abi
from citylearn_amd.engine import StepEngine as _StepEngine

pytestmark = pytest.mark.gpu


def StepEngine(*args, f64_maps=False, **kw):
    """This module pins LAUNCH GEOMETRY -- kernel template, envs per lane, chunking, deferred folds -- and most of those rules were measured
    on the all-fp32 battery map's kernels, so engines here are built with `f64_maps=False` unless a test says otherwise.  The default precision
    model (`CLD_F64_CHAIN` since round 6) has its own cases: `test_config_sizes_with_distinct_actions_against_the_c_oracle[..-default]`,
    `test_kernel_selection_by_batch_size[..-default]`, tests/test_gpu_parity.py."""
    return _StepEngine(*args, f64_maps=f64_maps, **kw)

REWARDS = ('RewardFunction', 'MARL', 'IndependentSACReward', 'SolarPenaltyReward')


def _err(got, ref, atol, rtol):
    ref = np.asarray(ref, dtype=np.float64)
    return float(np.max(np.abs(np.asarray(got, dtype=np.float64) - ref) / (atol + rtol * np.abs(ref))))


@pytest.mark.parametrize('kind', ['RewardFunction', 'MARL'])
@pytest.mark.parametrize('name,detail', [('g2023_p2', False), ('g2020_cz1', False), ('g2023_p2', True)])
def test_thermal_districts_at_65536_envs(name, kind, detail):
    """(1) replication: the 65 536-env batch is 512 distinct action columns tiled along the env axis, so env e must equal env
    e mod 512 of a 512-env engine stepped with the same actions, bit for bit on every state / net / reward (/ detail) plane;
    (2) the district sums are the sums of the building planes; (3) a checksum of checksums over the whole batch."""
    g = golden(name)
    spec = g.spec()
    tab = spec.episode_tables(0)
    E, S = 65536, 512
    # at 65 536 envs districts of 6 .. 16 buildings run cl_step_full_tp_kernel (two env tiles per workgroup, district sums in building
    # order); the 512-env engine is put on the same kernel so that the district sums can be compared bit for bit as well
    multi_tile = not detail and 6 <= len(spec.buildings) <= 16
    small = StepEngine(tab, S, reward=kind, detail=detail, tuning=dict(full_variant=5) if multi_tile else None)
    big = StepEngine(tab, E, reward=kind, detail=detail)
    assert not small.lean
    low, high = spec.action_limits()
    lo, hi = torch.from_numpy(low).cuda()[:, None], torch.from_numpy(high).cuda()[:, None]
    gen = torch.Generator(device='cuda').manual_seed(len(name) + len(kind))
    reps = E // S
    for t in range(30):
        a = (lo + torch.rand((small.n_act_cols, S), device='cuda', generator=gen) * (hi - lo)).contiguous()
        a[:, 0] = 0.0
        a[:, 1], a[:, 2] = lo[:, 0], hi[:, 0]
        small.step(a, t)
        big.step(a.repeat(1, reps).contiguous(), t)
    torch.cuda.synchronize()
    planes = [('state', small.state, big.state), ('net', small.out_bldg[abi.CLO_NET], big.out_bldg[abi.CLO_NET]),
              ('reward', small.out_bldg[abi.CLO_REWARD], big.out_bldg[abi.CLO_REWARD])]
    if detail:
        planes.append(('detail', small.out_bldg[abi.CLO_B_EB:abi.CLO_RESERVED], big.out_bldg[abi.CLO_B_EB:abi.CLO_RESERVED]))
    for label, s_plane, b_plane in planes:
        tiled = b_plane.reshape(*b_plane.shape[:-1], reps, S)
        assert torch.equal(tiled, s_plane.unsqueeze(-2).expand_as(tiled)), label
    tiled = big.out_env.reshape(abi.CL_NQ, reps, S)
    assert torch.equal(tiled, small.out_env.unsqueeze(1).expand_as(tiled))            # same workgroup shape -> same summation order
    net_sum = big.out_bldg[abi.CLO_NET].double().sum(dim=0)
    torch.testing.assert_close(big.out_env[abi.CLQ_NET].double(), net_sum, rtol=1e-6, atol=1e-4)
    torch.testing.assert_close(big.out_env[abi.CLQ_REWARD].double(), big.out_bldg[abi.CLO_REWARD].double().sum(dim=0), rtol=1e-6, atol=1e-3)
    total_small = small.out_bldg[abi.CLO_NET].double().sum().item()
    assert abs(big.out_bldg[abi.CLO_NET].double().sum().item() - reps * total_small) <= 1e-9 * abs(reps * total_small) + 1e-6
    assert small.state.abs().sum().item() > 0 and small.out_bldg[abi.CLO_NET].abs().sum().item() > 0


@pytest.mark.parametrize('precision', ['default', 'fp32'])
@pytest.mark.parametrize('name,E,steps,kind', [('g2022_all', 65536, 6, 'RewardFunction'), ('g2022_all', 65536, 4, 'MARL'), ('g2023_p2', 65536, 6, 'RewardFunction'),
                                               ('g2020_cz1', 65536, 4, 'SolarPenaltyReward'), ('g2022_all', 262144, 3, 'RewardFunction')])
def test_config_sizes_with_distinct_actions_against_the_c_oracle(name, E, steps, kind, precision):
    """at the config sizes the replication tests above compare the engine with itself; here EVERY env of the headline shape
    (17 x 65 536, the lean kernel at four envs per lane), of C3's energy step (2023 schema, 3 x 65 536), of the thermal 9 x 65 536 shape
    (multi-tile kernel) and of the 17 x 262 144 shape (env-major kernel) has its own actions (bounds and zeros included) and is compared
    with the C restatement of the reference arithmetic (oracle/cl_oracle.c, float64, OpenMP over envs) on every state plane, net,
    reward and district sum -- teacher-forced from the oracle's state each step, rows inside the 2023 fixture's outage included."""
    from oracle.c_oracle import COracle, OS, OO
    g = golden(name)
    spec = g.spec()
    tab = spec.episode_tables(0)
    eng = StepEngine(tab, E, reward=kind, f64_maps=None if precision == 'default' else False)
    assert eng.f64_chain == (precision == 'default')
    eng.trace_kernels()
    ora = COracle(spec, tab, E, reward=kind)
    low, high = spec.action_limits()
    rng = np.random.RandomState(E % 1000 + len(name))
    t0 = 388 if name == 'g2023_p2' else 0                    # (the fixture's power outage covers rows 389 - 403)
    worst = {}
    planes = ((abi.CLS_B_SOC, 'SOC'), (abi.CLS_B_EFF, 'EFF'), (abi.CLS_B_DEGCAP, 'DEGCAP'), (abi.CLS_CS_SOC, 'CS'), (abi.CLS_HS_SOC, 'HS'), (abi.CLS_DS_SOC, 'DS'))
    if t0:                                                   # mid-episode start: some charge in every storage
        ora.state[:, :, OS['SOC']] = rng.uniform(0.2, 0.8, size=ora.state.shape[:2])
        ora.state[:, :, OS['DS']] = rng.uniform(0.0, 0.6, size=ora.state.shape[:2])
    for t in range(t0, t0 + steps):
        a = rng.uniform(low[:, None], high[:, None], size=(len(low), E)).astype(np.float32)
        a[:, 0] = 0.0
        a[:, 1], a[:, 2] = low, high
        for pl, key in planes:
            eng.state[pl] = torch.from_numpy(np.ascontiguousarray(ora.state[:, :, OS[key]].T).astype(np.float32)).cuda()
        if eng.f64_chain:                                    # (the plane carries the capacity LOSS under the float64 chain)
            eng.state[abi.CLS_B_DEGCAP] = eng.params[:, abi.CLP_L_CAP].view(torch.float32)[:, None] - eng.state[abi.CLS_B_DEGCAP]
        eng.step(torch.from_numpy(a).cuda(), t)
        out, oe = ora.step(a, t)
        got_state = {pl: (eng.degraded_capacity if pl == abi.CLS_B_DEGCAP else eng.state[pl]).cpu().numpy() for pl, _ in planes}
        checks = [(key.lower(), got_state[pl], ora.state[:, :, OS[key]].T, 1e-4, 1e-4) for pl, key in planes]
        checks += [('net', eng.net.cpu().numpy(), out[:, :, OO['NET']].T, 1e-4, 1e-4),
                   ('reward', eng.reward_bldg.cpu().numpy(), out[:, :, OO['REWARD']].T, 1e-4, 1e-4),
                   ('d_net', eng.district_net.cpu().numpy(), oe[:, 0], 1e-4, 1e-4), ('d_cost', eng.out_env[abi.CLQ_COST].cpu().numpy(), oe[:, 1], 1e-4, 1e-4),
                   ('d_emission', eng.out_env[abi.CLQ_EMISSION].cpu().numpy(), oe[:, 2], 1e-4, 1e-4),
                   ('d_reward', eng.district_reward.cpu().numpy(), oe[:, 3], 1e-4, 1e-4)]
        for key, got, ref, atol, rtol in checks:
            worst[key] = max(worst.get(key, 0.0), _err(got, ref, atol, rtol))
    expect = {('g2022_all', 65536): 'cl_step_lean_kernel<4', ('g2023_p2', 65536): 'cl_step_full', ('g2020_cz1', 65536): 'cl_step_full_tp_kernel',
              ('g2022_all', 262144): 'cl_step_envmajor_kernel'}[(name, E)]
    if eng.f64_chain:
        assert 'chain' in eng.last_kernels or E == 262144, eng.last_kernels
        expect = {('g2022_all', 65536): 'cl_step_lean_chain_kernel<4', ('g2023_p2', 65536): 'cl_step_full_', ('g2020_cz1', 65536): 'cl_step_full_tp_chain_kernel',      # (3 x 65 536 at one env per lane: four tiles per workgroup -> the multi-tile kernel too)
                  ('g2022_all', 262144): 'cl_step_envmajor_kernel<17, '}[(name, E)]
        assert E != 262144 or eng.last_kernels.endswith(', 1, 2>'), eng.last_kernels
    assert expect in eng.last_kernels, eng.last_kernels
    check_worst(worst, f'{name} x {E} {kind} {precision}')


@pytest.mark.parametrize('E,expect', [(16384, 'cl_step_lean_kernel<1, '), (32768, 'cl_step_lean_kernel<2, '), (65536, 'cl_step_lean_kernel<4, '),
                                       (98304, 'cl_step_lean_kernel<4, '), (122880, 'cl_step_lean_kernel<4, '), (122884, 'cl_step_envmajor_kernel<17, '),
                                       (196608, 'cl_step_envmajor_kernel<17, '), (196612, 'cl_step_envmajor_kernel<17, '), (262144, 'cl_step_envmajor_kernel<17, '),
                                       (524288, 'cl_step_lean_kernel<4, false, true>'), (1048576, 'cl_step_lean_kernel<4, ')])
@pytest.mark.parametrize('precision', ['default', 'fp32'])
def test_kernel_selection_by_batch_size(E, expect, precision):
    """Which kernel steps the 17-building battery + PV district at which batch size (csrc/cl_kernels.hip step_impl; re-measured at the end of
    round 5, profiles/r05_nt_loads/r05y.log, and in round 6 with both kernels alternating in one process, profiles/r06_lean_vs_envmajor*.log): the
    latency-ordered kernel at one / two / four envs per lane while the launch is one wave generation (up to 480 workgroups = 122 880 envs; under the
    float64 chain up to 196 608 envs), the env-major kernel beyond -- and the latency-ordered kernel again from 8 Mi units (17 x 1 048 576, the HBM-true
    shape of the bench line: 16-byte accesses win far beyond the Infinity Cache).  The kernels agree on every per-building plane."""
    tab = golden('g2022_all').spec().episode_tables(0)
    f64 = None if precision == 'default' else False
    eng = StepEngine(tab, E, f64_maps=f64)
    eng.trace_kernels()
    ref = StepEngine(tab, E, f64_maps=f64, tuning=dict(lean_variant=1, envmajor=2))          # the general kernel
    if eng.f64_chain:                                  # (the chain instantiations; the env-major kernel takes over later: csrc/cl_kernels.hip step_impl)
        if E in (122884, 196608):
            expect = 'cl_step_lean_kernel<4, '
        expect = expect.replace('cl_step_lean_kernel<4, false, true>', 'cl_step_lean_kernel<4, true>').replace('cl_step_lean_kernel<', 'cl_step_lean_chain_kernel<')
    gen = torch.Generator(device='cuda').manual_seed(E)
    for t in range(3):
        a = torch.rand((eng.n_act_cols, E), device='cuda', generator=gen) * 2 - 1
        eng.step(a, t); ref.step(a, t)
    assert eng.last_kernels.startswith(expect), eng.last_kernels
    assert torch.equal(eng.state, ref.state) and torch.equal(eng.out_bldg[:2], ref.out_bldg[:2])
    torch.testing.assert_close(eng.out_env, ref.out_env, rtol=2e-6, atol=2e-5)


@lru_cache(maxsize=None)
def _c4_district(fixture: str):
    from citylearn_amd.synthetic import tile_district
    spec = tile_district(golden(fixture).spec(), 1024)
    return spec, spec.episode_tables(0)


@pytest.mark.parametrize('kind', REWARDS)
@pytest.mark.parametrize('fixture', ['g2020_cz1', 'g2022_all'])
def test_c4_shard_1024_buildings_x_1024_envs(fixture, kind):
    """BASELINE config 4's per-GPU shard (synthetic district: the fixture's buildings tiled to 1024 with jittered device sizes,
    1024 envs): 64 building chunks along gridDim.y, partial district sums finished by cl_finish_kernel, MARL's per-building
    reward by cl_marl_reward_kernel.  Every env has its own actions; each step starts from the C oracle's state."""
    _c4_against_the_c_oracle(fixture, kind, 1024, 12)


def _c4_against_the_c_oracle(fixture, kind, E, steps, tuning=None, f64_maps=False):
    from oracle.c_oracle import COracle, OS, OO
    spec, tab = _c4_district(fixture)
    eng = StepEngine(tab, E, reward=kind, tuning=tuning, f64_maps=f64_maps)
    ora = COracle(spec, tab, E, reward=kind)
    assert eng.n_bldg == 1024
    low, high = spec.action_limits()
    rng = np.random.RandomState(11)
    worst = {}
    for t in range(steps):
        a = rng.uniform(low[:, None], high[:, None], size=(len(low), E)).astype(np.float32)
        a[:, 0] = 0.0
        a[:, 1], a[:, 2] = low, high
        for pl, key in ((abi.CLS_B_SOC, 'SOC'), (abi.CLS_B_EFF, 'EFF'), (abi.CLS_B_DEGCAP, 'DEGCAP'), (abi.CLS_CS_SOC, 'CS'),
                        (abi.CLS_HS_SOC, 'HS'), (abi.CLS_DS_SOC, 'DS')):
            eng.state[pl] = torch.from_numpy(np.ascontiguousarray(ora.state[:, :, OS[key]].T).astype(np.float32)).cuda()
        if eng.f64_chain:                                    # (the plane carries the capacity LOSS under the float64 chain)
            eng.state[abi.CLS_B_DEGCAP] = eng.params[:, abi.CLP_L_CAP].view(torch.float32)[:, None] - eng.state[abi.CLS_B_DEGCAP]
        eng.step(torch.from_numpy(a).cuda(), t)
        out, oe = ora.step(a, t)
        got_net, got_rw = eng.net.cpu().numpy(), eng.reward_bldg.cpu().numpy()
        # per-building reward: the plain bar -- except MARL / SolarPenaltyReward, steep functions of `net` (see coupled_reward_tolerance: the one
        # place of the suite where the bar on a reward is the propagated one; `reward_plain` records what the plain bar would read)
        ref_rw = out[:, :, OO['REWARD']].T
        f = tab.params[:, abi.CLP_FLAGS]
        n_sto = sum(((f & bit) != 0).astype(np.float64) for bit in (abi.CLF_BATTERY, abi.CLF_COOL_STO, abi.CLF_HEAT_STO, abi.CLF_DHW_STO))
        rw_tol = coupled_reward_tolerance(kind, ref_rw, out[:, :, OO['NET']].T, oe[:, 0], n_sto)
        worst['reward_plain'] = max(worst.get('reward_plain', 0.0), _err(got_rw, ref_rw, 1e-4, 1e-4))
        # district sums over 1024 buildings are O(1e3 kWh) in fp32: absolute tolerance scaled with the district size; the per-building
        # reward tolerance is the one of test_large_district_building_chunked_grid (SolarPenaltyReward multiplies |net| by four SoCs)
        for key, e in (('soc', _err(eng.soc.cpu().numpy(), ora.state[:, :, OS['SOC']].T, 1e-4, 1e-4)),
                       ('ds_soc', _err(eng.state[abi.CLS_DS_SOC].cpu().numpy(), ora.state[:, :, OS['DS']].T, 1e-4, 1e-4)),
                       ('net', _err(got_net, out[:, :, OO['NET']].T, 1e-4, 1e-4)),
                       ('reward', float(np.max(np.abs(got_rw.astype(np.float64) - ref_rw) / rw_tol))),
                       ('d_net', _err(eng.district_net.cpu().numpy(), oe[:, 0], 1e-4, 1e-4)),
                       ('d_cost', _err(eng.out_env[abi.CLQ_COST].cpu().numpy(), oe[:, 1], 1e-4, 1e-4)),
                       ('d_emission', _err(eng.out_env[abi.CLQ_EMISSION].cpu().numpy(), oe[:, 2], 1e-4, 1e-4)),
                       ('d_reward', _err(eng.district_reward.cpu().numpy(), oe[:, 3], 1e-4, 1e-4))):
            worst[key] = max(worst.get(key, 0.0), e)
        # the finished sums are the sums of the planes the chunks wrote
        torch.testing.assert_close(eng.district_net.double(), eng.net.double().sum(dim=0), rtol=1e-5, atol=1e-2)
        torch.testing.assert_close(eng.district_reward.double(), eng.reward_bldg.double().sum(dim=0), rtol=2e-5, atol=1e-2)
    plain = worst.pop('reward_plain')
    check_worst({**worst, **({'reward_plain': plain} if kind not in ('MARL', 'SolarPenaltyReward') else {})}, f'C4 {fixture} x {E} {kind}')
    print(f'C4 {fixture} x {E} {kind}: per-building reward at the plain bar {plain:.3f}, at the gate {worst["reward"]:.3f}')
    return eng


@pytest.mark.parametrize('fixture', ['g2020_cz1', 'g2022_all'])
def test_c4_shard_under_the_default_precision_model(fixture):
    """... and the same shard as `StepEngine` steps it by default since round 6: the battery's soc chain in float64 (CLD_F64_CHAIN) inside the
    building-chunked launches, both device sets, against the C oracle at the plain bar."""
    eng = _c4_against_the_c_oracle(fixture, 'RewardFunction', 1024, 8, f64_maps=None)
    assert eng.f64_chain


@pytest.mark.parametrize('fixture', ['g2020_cz1', 'g2022_all'])
def test_c4_whole_config_1024_buildings_x_8192_envs(fixture):
    """BASELINE config 4 WHOLE on one GPU -- 8.4 M (env, building) units, every env with its own actions, each step from the C oracle's
    state -- under the tuning `bench.py --config C4 --envs-per-gpu 8192` runs with (deferred finish): the thermal district in the chunk
    geometry the library picks for multi-generation batches (8 chunks of 128 buildings, 64 district sums folded per workgroup row), the
    battery + PV district in chunks of 32 with the second launch."""
    eng = _c4_against_the_c_oracle(fixture, 'RewardFunction', 8192, 3, tuning=dict(finish=3))
    eng.trace_kernels()
    a = torch.zeros((eng.n_act_cols, 8192), device='cuda')
    eng.step(a, 3)
    if fixture == 'g2020_cz1':
        assert eng.last_kernels == 'cl_step_full_kernel<2, false, 1024, 4, true, true>', eng.last_kernels      # folds its predecessor's sums: no second launch
    else:
        assert eng.last_kernels == 'cl_step_lean_chunk_kernel<4, false, false, 0>+cl_finish_kernel', eng.last_kernels


@pytest.mark.parametrize('kind', REWARDS)
@pytest.mark.parametrize('fixture', ['g2020_cz1', 'g2022_all'])
def test_c4_shard_deferred_finish(fixture, kind):
    """`tuning={'finish': 3}` (round 4): the step launch folds the PREVIOUS step's chunk partial sums and leaves its own to the next
    launch / to `cl_finish_f32` -- no second launch per step.  Free-running next to the default engine (second launch every step) on
    the C4 per-GPU shard, both device sets, all four rewards (MARL couples the buildings and keeps its second launch: identical
    trivially): every plane and -- once folded -- every district sum bit for bit; before the fold `out_env` trails by exactly one
    step; `cl_finish_f32` is idempotent and a no-op for engines that did not defer; `step_many` finishes its last step itself."""
    spec, tab = _c4_district(fixture)
    E, K = 1024, 6
    ref = StepEngine(tab, E, reward=kind)
    dfr = StepEngine(tab, E, reward=kind, tuning=dict(finish=3))
    ref.trace_kernels(); dfr.trace_kernels()
    low, high = spec.action_limits()
    rng = np.random.RandomState(3)
    acts = torch.from_numpy(rng.uniform(low[:, None], high[:, None], size=(K, len(low), E)).astype(np.float32)).cuda()
    prev = None
    for t in range(K):
        ref.step(acts[t], t)
        dfr.step(acts[t], t)
        if kind == 'MARL':
            assert 'cl_finish_kernel' in dfr.last_kernels and dfr.last_kernels == ref.last_kernels
        else:
            assert 'cl_finish_kernel' in ref.last_kernels and 'cl_finish_kernel' not in dfr.last_kernels, dfr.last_kernels
            if prev is not None:
                assert torch.equal(dfr._out_env, prev)                # not folded yet: the previous step's district sums
        assert torch.equal(dfr.state, ref.state)
        assert torch.equal(dfr.out_bldg[:abi.CLO_RESERVED], ref.out_bldg[:abi.CLO_RESERVED])
        if t % 2 == 0 or t == K - 1:                                  # (odd steps are folded by the next launch instead)
            assert torch.equal(dfr.out_env, ref.out_env), (t, (dfr.out_env - ref.out_env).abs().max().item())
            dfr._pending_t = t
            assert torch.equal(dfr.out_env, ref.out_env)              # a second cl_finish_f32 of the same step changes nothing
        prev = ref.out_env.clone()
    # an engine that does not defer: cl_finish_f32 finds no marker and leaves out_env alone
    before = ref._out_env.clone()
    ref._pending_t = K - 1
    ref.finish()
    assert torch.equal(ref._out_env, before)
    # a step that does NOT defer clears its parity's marker: switching the mode on live buffers cannot make a later cl_finish_f32 fold a stale buffer
    if kind != 'MARL':
        dfr.step(acts[1], 1)                                           # deferred: marker of parity 1 set, out_env not folded yet
        dfr.tuning.finish = 1                                          # (the engine reads the mode from the tuning block at every call)
        dfr.step(acts[1], 1)                                           # the same step again with the second launch
        after = dfr._out_env.clone()
        dfr._pending_t = 1
        dfr.finish()
        assert torch.equal(dfr._out_env, after)
        dfr.tuning.finish = 3
    # step_many (cl_rollout_seq_f32) folds its last step itself
    ref.reset(); dfr.reset()
    ref.step_many(acts); dfr.step_many(acts)
    assert dfr._pending_t is None and torch.equal(dfr._out_env, ref._out_env) and torch.equal(dfr.state, ref.state)
    ret_r, ret_d = torch.zeros(E, device='cuda'), torch.zeros(E, device='cuda')
    ref.reset(); dfr.reset()
    ref.rollout(K, actions=acts, ret_env=ret_r); dfr.rollout(K, actions=acts, ret_env=ret_d)
    assert torch.equal(ret_d, ret_r) and torch.equal(dfr._out_env, ref._out_env)


@pytest.mark.parametrize('fixture,b_chunk,E', [('g2020_cz1', 128, 1024), ('g2020_cz1', 64, 640), ('g2020_cz1', None, 4096)])
def test_deferred_finish_with_larger_chunks(fixture, b_chunk, E):
    """Round 5: the deferred fold's exchange tile holds [chunks][16 / 32 / 64 district sums], so launches with fewer, larger chunks defer
    too -- 8 chunks of 128 buildings (64 sums per workgroup row, four per wave), 16 of 64 (32 sums; ragged last env tile), and the
    geometry the library now picks itself for the thermal district at 4096 envs (`b_chunk=None`: chunks of 128, 256 workgroups).
    Planes and -- once folded -- district sums bit for bit against the same geometry with the second launch.  (Thermal districts only:
    the battery + PV kernel keeps the second launch where a row would fold more than 16 sums -- measured slower.)"""
    spec, tab = _c4_district(fixture)
    K = 5
    geo = dict(b_chunk=b_chunk) if b_chunk else {}
    ref = StepEngine(tab, E, tuning=dict(finish=1, **geo))
    dfr = StepEngine(tab, E, tuning=dict(finish=3, **geo))
    ref.trace_kernels(); dfr.trace_kernels()
    low, high = spec.action_limits()
    rng = np.random.RandomState(E)
    acts = torch.from_numpy(rng.uniform(low[:, None], high[:, None], size=(K, len(low), E)).astype(np.float32)).cuda()
    prev = None
    for t in range(K):
        ref.step(acts[t], t); dfr.step(acts[t], t)
        assert 'cl_finish_kernel' in ref.last_kernels and 'cl_finish_kernel' not in dfr.last_kernels, (ref.last_kernels, dfr.last_kernels)
        if prev is not None:
            assert torch.equal(dfr._out_env, prev)                    # not folded yet: the previous step's district sums
        assert torch.equal(dfr.state, ref.state)
        assert torch.equal(dfr.out_bldg[:abi.CLO_RESERVED], ref.out_bldg[:abi.CLO_RESERVED])
        if t % 2 == 0 or t == K - 1:
            assert torch.equal(dfr.out_env, ref.out_env), (t, (dfr.out_env - ref.out_env).abs().max().item())
        prev = ref.out_env.clone()
    torch.testing.assert_close(ref.district_net.double(), ref.net.double().sum(dim=0), rtol=1e-5, atol=1e-2)
    assert float(ref.out_env.abs().sum()) > 0
    ref.reset(); dfr.reset()
    ref.step_many(acts); dfr.step_many(acts)
    assert dfr._pending_t is None and torch.equal(dfr._out_env, ref._out_env) and torch.equal(dfr.state, ref.state)


@pytest.mark.parametrize('f64', [False, 'chain'])
@pytest.mark.parametrize('E,kind,finish', [(1024, 'RewardFunction', 3), (1024, 'MARL', 0), (1280, 'SolarPenaltyReward', 0), (256, 'RewardFunction', 0),
                                            (8192, 'RewardFunction', 3)])
def test_lean_chunk_kernel_equals_the_general_kernel(E, kind, finish, f64):
    """`cl_step_lean_chunk_kernel` (round 5): the building-chunked battery + PV launch with the next building's plane loads issued ahead of
    the current one's stores -- same arithmetic and summation order as `cl_step_kernel<VEC, false, false>` (`lean_variant = 16`), so every
    plane and every district sum bit for bit: four envs per lane (1024 / 1280 envs: ragged last tile; 8192: the whole config), one env per
    lane (256 envs), the deferred fold, MARL's chunk-partial reward."""
    spec, tab = _c4_district('g2022_all')
    K = 4
    # (round 6: also around the float64 soc chain -- CLD_F64_CHAIN, the engine's default precision model -- with the same deferred fold)
    new = StepEngine(tab, E, reward=kind, tuning=dict(finish=finish), f64_maps=f64)
    old = StepEngine(tab, E, reward=kind, tuning=dict(finish=finish, lean_variant=16), f64_maps=f64)
    new.trace_kernels(); old.trace_kernels()
    low, high = spec.action_limits()
    rng = np.random.RandomState(E)
    acts = torch.from_numpy(rng.uniform(low[:, None], high[:, None], size=(K, len(low), E)).astype(np.float32)).cuda()
    for t in range(K):
        new.step(acts[t], t); old.step(acts[t], t)
        assert new.last_kernels.startswith('cl_step_lean_chunk_kernel<') and 'lean_chunk' not in old.last_kernels, (new.last_kernels, old.last_kernels)
        assert torch.equal(new.state, old.state)
        assert torch.equal(new.out_bldg[:abi.CLO_RESERVED], old.out_bldg[:abi.CLO_RESERVED])
        assert torch.equal(new.out_env, old.out_env), (t, (new.out_env - old.out_env).abs().max().item())
    vec = 4 if E >= 512 else 1
    deferred = finish == 3 and kind != 'MARL' and E <= 1280
    assert new.last_kernels.split('+')[0] == f'cl_step_lean_chunk_kernel<{vec}, true, {"true" if deferred else "false"}, {2 if f64 else 0}>' or E == 8192, new.last_kernels
    assert float(new.out_env.abs().sum()) > 0


def test_lean_chunk_kernel_with_a_sparse_action_map():
    """... and where the battery action column of building b is NOT b (two of 64 buildings without an active battery action: the flag
    CLD_ES_COL_IS_BLDG is off, the kernel loads the action once the parameter block is there) -- same bits as the general kernel."""
    from dataclasses import replace
    from citylearn_amd.synthetic import tile_district
    spec = tile_district(golden('g2022_all').spec(), 64)
    blds = list(spec.buildings)
    for i in (3, 10):
        blds[i].action_metadata = {**blds[i].action_metadata, 'electrical_storage': False}
    spec = replace(spec, buildings=blds)
    tab = spec.episode_tables(0)
    E = 8192
    new, old = StepEngine(tab, E), StepEngine(tab, E, tuning=dict(lean_variant=16))
    assert new.n_act_cols == 62 and not (new.dims.flags & abi.CLD_ES_COL_IS_BLDG)
    new.trace_kernels()
    gen = torch.Generator(device='cuda').manual_seed(5)
    for t in range(4):
        a = torch.rand((new.n_act_cols, E), device='cuda', generator=gen) * 2 - 1
        new.step(a, t); old.step(a, t)
        assert torch.equal(new.state, old.state) and torch.equal(new.out_bldg[:2], old.out_bldg[:2]) and torch.equal(new.out_env, old.out_env), t
    assert new.last_kernels.startswith('cl_step_lean_chunk_kernel<4, '), new.last_kernels
    assert float(new.state[abi.CLS_B_SOC, 3].abs().max()) == 0.0 and float(new.state[abi.CLS_B_SOC, 4].abs().max()) > 0.0      # the idle battery stays empty


def test_deferred_finish_through_step_observe():
    """`StepEngine.step_observe`'s one-call path (`cl_step_observe_f32`) runs the same step launch as `step` -- on a chunked
    district under `finish = 3` it defers the district sums too, so a later read of `out_env` must fold them (it used to return the
    previous step's).  A 256-building battery + PV district of which only the first eight buildings expose env-dependent observations
    (the compact form holds at most 64 columns), 1024 envs: 16 chunks, deferred fold."""
    import copy
    from dataclasses import replace
    from citylearn_amd.observations import ObservationLayout
    from citylearn_amd.observe import ObservationWriter
    from citylearn_amd.synthetic import tile_district
    spec = tile_district(golden('g2022_all').spec(), 256)
    blds = []
    for i, b in enumerate(spec.buildings):
        b = copy.copy(b)
        if i >= 8:
            b.observation_metadata = {k: (k == 'hour') for k in b.observation_metadata}
        blds.append(b)
    spec = replace(spec, buildings=blds)
    tab = spec.episode_tables(0)
    dep_tables, _ = ObservationLayout(spec, 'current', False).episode(tab).compact()
    E = 1024
    ref, dfr = StepEngine(tab, E), StepEngine(tab, E, tuning=dict(finish=3))
    wa, wb = ObservationWriter(ref, dep_tables, None), ObservationWriter(dfr, dep_tables, None)
    assert 0 < wa.n_deps == wa.n_cols <= 64
    dfr.trace_kernels()
    gen = torch.Generator(device='cuda').manual_seed(9)
    for t in range(5):
        act = torch.rand((ref.n_act_cols, E), device='cuda', generator=gen) * 2 - 1
        ref.step(act, t)
        want = wa.write(t + 1).clone()
        got = dfr.step_observe(act, wb, t)
        assert 'cl_finish_kernel' not in dfr.last_kernels and dfr.last_kernels.endswith(', true, 0>'), dfr.last_kernels      # the FOLD instantiation (fp32 map), deferred
        assert dfr._pending_t == t
        assert torch.equal(got, want) and torch.equal(dfr.state, ref.state), t
        if t % 2 == 0:                                                  # (odd steps are folded by the next launch instead)
            assert torch.equal(dfr.out_env, ref.out_env), t
            assert dfr._pending_t is None
    assert torch.equal(dfr.out_env, ref.out_env)


@pytest.mark.parametrize('kind', ['RewardFunction', 'MARL'])
def test_c5_rollout_kernel_at_131072_envs(kind):
    """BASELINE config 5's kernel template -- two envs per lane, two buildings per wave (`cl_rollout_kernel<2, false, 2>`,
    selected from 131 072 envs up) -- with the on-device Philox policy, against K calls of cl_step_f32 fed with the host
    definition's actions for a sample of envs and, for the whole batch, against a small rollout by replication of the
    counter space (env index is part of the Philox counter, so no two envs share actions: compare with single steps instead)."""
    g = golden('g2022_all')
    spec = g.spec()
    tab = spec.episode_tables(0)
    E, K, seed = 131072, 24, 5
    low, high = spec.action_limits()
    roll = StepEngine(tab, E, reward=kind)
    roll.set_action_limits(low, high)
    ret = torch.zeros(E, device='cuda')
    roll.rollout(K, seed=seed, ret_env=ret)
    # the same K steps as single launches, actions regenerated from the documented stream by a second rollout engine forced to
    # one env per lane (the VEC = 1 kernel is the one tests/test_gpu_rollout.py pins on cl_step_f32 and the host Philox)
    one = StepEngine(tab, E, reward=kind, tuning=dict(vec=1))
    one.set_action_limits(low, high)
    ret1 = torch.zeros(E, device='cuda')
    one.rollout(K, seed=seed, ret_env=ret1)
    assert torch.equal(roll.state, one.state)
    assert torch.equal(roll.out_bldg[:2], one.out_bldg[:2])
    # district cost / emission are running sums of per-lane products, which the compiler may or may not fuse into the
    # accumulation per instantiation: last-bit differences only
    torch.testing.assert_close(roll.out_env, one.out_env, rtol=2e-6, atol=2e-5)
    torch.testing.assert_close(ret, ret1, rtol=2e-6, atol=1e-4)
    # and directly against cl_step_f32 on the tail of the batch (last 256 envs: the highest counters / addresses)
    from citylearn_amd import _lib
    lib = _lib.load()
    n = 256
    envs = np.arange(E - n, E)
    u = np.array([[[lib.cl_philox_uniform(seed, int(e), c, t) for e in envs] for c in range(len(low))] for t in range(K)], dtype=np.float32)
    acts = torch.from_numpy((low[None, :, None] + u * (high - low)[None, :, None]).astype(np.float32)).cuda()
    step = StepEngine(tab, n, reward=kind)
    ret_ref = torch.zeros(n, device='cuda')
    for k in range(K):
        step.step(acts[k])
        ret_ref += step.district_reward
    torch.testing.assert_close(roll.state[:, :, E - n:], step.state, rtol=2e-6, atol=2e-6)
    torch.testing.assert_close(roll.out_bldg[:2, :, E - n:], step.out_bldg[:2], rtol=2e-5, atol=2e-5)
    torch.testing.assert_close(ret[E - n:], ret_ref, rtol=1e-5, atol=1e-3)


@pytest.mark.parametrize('name,E,kind', [('g2022_all', 65536, 'RewardFunction'), ('g2022_all', 131072, 'MARL'), ('g2020_cz1', 16384, 'SolarPenaltyReward'),
                                         ('g2023_p2', 65536, 'RewardFunction')])
def test_store_policy_does_not_change_results(name, E, kind):
    """Non-temporal plane stores / loads (`cl_tuning.nt_stores`, selected by launch size: csrc/cl_kernels.hip `pstore`) are a cache hint:
    every plane and district sum is bit-identical with the hint forced on and forced off -- lean kernel at the headline size, env-major
    kernel, thermal kernel (one and nine buildings per district row)."""
    tab = golden(name).spec().episode_tables(0)
    on, off = (StepEngine(tab, E, reward=kind, tuning=dict(nt_stores=v)) for v in (1, 2))
    gen = torch.Generator(device='cuda').manual_seed(11)
    for t in range(6):
        a = torch.rand((on.n_act_cols, E), device='cuda', generator=gen) * 2 - 1
        on.step(a, t); off.step(a, t)
        assert torch.equal(on.state, off.state) and torch.equal(on.out_env, off.out_env), t
        assert torch.equal(on.out_bldg[:2], off.out_bldg[:2]), t


@pytest.mark.parametrize('kind', ['RewardFunction', 'MARL'])
def test_lds_staged_parameters_match_scalar_parameters(kind):
    """Building-chunked launches read their parameter blocks from LDS (`cl_step_full_kernel<.., LP = true>`); `full_variant = 3` keeps
    them in SGPRs.  Same arithmetic, same bits -- on a 200-building thermal district (13 chunks of 16 buildings, ragged last chunk,
    ragged env tile) and with the staging forced on a district that is not chunked."""
    from citylearn_amd.synthetic import tile_district
    spec = tile_district(golden('g2020_cz1').spec(), 200)
    tab = spec.episode_tables(0)
    low, high = spec.action_limits()
    lo, hi = torch.from_numpy(low).cuda()[:, None], torch.from_numpy(high).cuda()[:, None]
    for tables, E, tun_lds in ((tab, 388, dict()), (golden('g2020_cz1').spec().episode_tables(0), 516, dict(full_variant=2))):
        lds, sgpr = StepEngine(tables, E, reward=kind, tuning=tun_lds), StepEngine(tables, E, reward=kind, tuning=dict(full_variant=3))
        n = lds.n_act_cols
        gen = torch.Generator(device='cuda').manual_seed(5)
        for t in range(8):
            u = torch.rand((n, E), device='cuda', generator=gen)
            a = (lo[:n] + u * (hi[:n] - lo[:n])).contiguous() if n == len(low) else u * 2 - 1
            lds.step(a, t); sgpr.step(a, t)
            assert torch.equal(lds.state, sgpr.state) and torch.equal(lds.out_env, sgpr.out_env), t
            assert torch.equal(lds.out_bldg[:2], sgpr.out_bldg[:2]), t


@pytest.mark.parametrize('E,tuning', [(65536, None), (516, None), (132, dict(vec=4, lean_variant=2))])
def test_kpi_accumulators_updated_by_the_lean_step_kernel(E, tuning):
    """`CLD_KPI` without the detail planes (battery + PV districts of up to 32 buildings): `cl_step_lean_kpi_kernel` updates every
    accumulator inside the step launch -- the four control sums per (env, building) and the control district series per env; the
    baseline sums, the expected energy and the baseline district series, which do not depend on the env in such a district, once per
    block of CL_ROW0_BLOCK envs at the block's first env.  Same values as the two passes over the detail planes (`cl_kpi_bldg_kernel`,
    `cl_kpi_env_kernel`), which the reference-pinned KPI tests of tests/test_env_gpu.py cover."""
    tab = golden('g2022_all').spec().episode_tables(0)
    fused = StepEngine(tab, E, kpi=True, tuning=tuning)
    two_pass = StepEngine(tab, E, kpi=True, detail=True)
    fused.trace_kernels()
    assert not (fused.dims.flags & abi.CLD_WRITE_DETAIL) and (two_pass.dims.flags & abi.CLD_WRITE_DETAIL)
    assert fused.kpi_shared_baseline and not two_pass.kpi_shared_baseline
    gen = torch.Generator(device='cuda').manual_seed(E)
    for t in range(30):
        a = torch.rand((fused.n_act_cols, E), device='cuda', generator=gen) * 2 - 1
        fused.step(a, t); two_pass.step(a, t)
    assert fused.last_kernels.startswith('cl_step_lean_kpi_kernel<') and '+' not in fused.last_kernels          # one launch per step
    assert torch.equal(fused.state, two_pass.state) and torch.equal(fused.out_bldg[:2], two_pass.out_bldg[:2])
    control = [abi.CLK_C_POS, abi.CLK_C_NET, abi.CLK_C_EMISSION, abi.CLK_C_COST]
    shared = [abi.CLK_B_POS, abi.CLK_B_NET, abi.CLK_B_EMISSION, abi.CLK_B_COST, abi.CLK_EXPECTED_ALL]
    lead = torch.arange(0, E, abi.CL_ROW0_BLOCK, device='cuda')
    others = torch.ones(E, dtype=torch.bool, device='cuda')
    others[lead] = False
    torch.testing.assert_close(fused.kpi_bldg[control], two_pass.kpi_bldg[control], rtol=2e-6, atol=1e-5)
    torch.testing.assert_close(fused.kpi_bldg[shared][:, :, lead], two_pass.kpi_bldg[shared][:, :, lead], rtol=2e-6, atol=1e-5)
    assert not fused.kpi_bldg[shared][:, :, others].any()                       # kept once per block: the other entries stay at reset
    n = abi.CLKE_PER_COND
    torch.testing.assert_close(fused.kpi_env[:n], two_pass.kpi_env[:n], rtol=2e-5, atol=1e-4)
    torch.testing.assert_close(fused.kpi_env[n:][:, lead], two_pass.kpi_env[n:][:, lead], rtol=2e-5, atol=1e-4)
    # what evaluate() makes of them is the same
    from citylearn_amd.kpi import finalize_streaming
    b1, d1 = finalize_streaming(fused.kpi_bldg, fused.kpi_env, 30, tab.n_steps, shared_baseline=True)
    b2, d2 = finalize_streaming(two_pass.kpi_bldg, two_pass.kpi_env, 30, tab.n_steps)
    for k in b2:
        torch.testing.assert_close(b1[k], b2[k], rtol=1e-5, atol=1e-6, equal_nan=True)
    for k in d2:
        torch.testing.assert_close(d1[k], d2[k], rtol=1e-4, atol=1e-5, equal_nan=True)
    assert fused.kpi_bldg.abs().sum().item() > 0


def _kpi_step_waves(n_env, n_bldg):
    """cl_step_full_kpi_kernel's waves per workgroup (csrc/cl_kernels.hip step_impl): as many as keep every wave of the launch resident."""
    return min(n_bldg, max(2, min(16, 4096 // -(-n_env // 64))))


@pytest.mark.parametrize('name,E', [('g2020_cz1', 65536), ('g2023_p2', 516), ('g2022_evs', 260), ('s_2023_p3', 1028)])
def test_streaming_kpis_in_one_pass_over_the_minimal_detail_planes(name, E):
    """Districts whose KPI baseline depends on the env (thermal, outage, EV), three ways, same bits:
    * default, thermal / outage districts: the step launch updates every accumulator itself (`cl_step_full_kpi_kernel`, no detail planes);
    * `cl_tuning.kpi_passes = 1` (and EV districts by default): the step writes only the detail planes another kernel reads
      (`CLD_DETAIL_MIN`: baseline, expected, served, delivered demands) and ONE launch (`cl_kpi_kernel`) updates the accumulators;
    * `kpi_passes = 2`: the step with all fifteen detail planes followed by the two passes of rounds 1 - 2."""
    g = golden(name)
    spec = g.spec()
    tab = spec.episode_tables(0)
    fused = StepEngine(tab, E, kpi=True)                                 # thermal: in the step launch; EV: detail 'min' by itself
    # (the in-step launch picks its own number of waves per workgroup -- all waves resident at once -- and the district sums are added
    #  per wave, then over waves: the same geometry for the other two, so that every sum compares bit for bit)
    nw = {} if fused.flex is not None else dict(nw=_kpi_step_waves(E, fused.n_bldg))
    new = StepEngine(tab, E, kpi=True, tuning=dict(kpi_passes=1, **nw))        # detail 'min' + cl_kpi_kernel
    old = StepEngine(tab, E, kpi=True, detail=True, tuning=dict(kpi_passes=2, **nw))
    for e in (fused, new, old):
        e.trace_kernels()
    in_step = fused.flex is None
    assert fused.detail == (False if in_step else 'min')
    assert new.detail == 'min' and (new.dims.flags & abi.CLD_DETAIL_MIN) and not (old.dims.flags & abi.CLD_DETAIL_MIN)
    low, high = spec.action_limits()
    lo, hi = torch.from_numpy(low).cuda()[:, None], torch.from_numpy(high).cuda()[:, None]
    gen = torch.Generator(device='cuda').manual_seed(E)
    for t in range(40):
        a = (lo + torch.rand((new.n_act_cols, E), device='cuda', generator=gen) * (hi - lo)).contiguous()
        fused.step(a, t); new.step(a, t); old.step(a, t)
    assert fused.last_kernels.startswith('cl_step_full_kpi_kernel<') == in_step and ('cl_kpi_kernel' in fused.last_kernels) != in_step, fused.last_kernels
    assert new.last_kernels.endswith('+cl_kpi_kernel') and old.last_kernels.endswith('cl_kpi_bldg_kernel+cl_kpi_env_kernel')
    assert torch.equal(new.state, old.state) and torch.equal(new.out_env, old.out_env)
    assert torch.equal(fused.state, old.state) and torch.equal(fused.out_env, old.out_env)
    for pl in (abi.CLO_NET, abi.CLO_REWARD, abi.CLO_BASE_NET, abi.CLO_EXPECTED, abi.CLO_SERVED, abi.CLO_COOL_DEM, abi.CLO_HEAT_DEM):
        assert torch.equal(new.out_bldg[pl], old.out_bldg[pl]), pl
    assert torch.equal(fused.out_bldg[:2], old.out_bldg[:2])
    if in_step:
        assert not fused.out_bldg[abi.CLO_BASE_NET].any()                                       # no detail plane written at all
    assert not new.out_bldg[abi.CLO_C_NSL].any() and old.out_bldg[abi.CLO_C_NSL].any()          # the other planes are left alone
    assert torch.equal(new.kpi_bldg, old.kpi_bldg) and torch.equal(new.kpi_env, old.kpi_env)
    assert torch.equal(fused.kpi_bldg, old.kpi_bldg) and torch.equal(fused.kpi_env, old.kpi_env)
    assert float(new.kpi_bldg.abs().sum()) > 0


def test_streaming_kpis_in_the_thermal_step_launch_with_detail_planes_and_episode_offsets():
    """`cl_step_full_kpi_kernel` also when the caller wants detail planes (all of them: observations / plugins; the subset: the LSTM
    stage) and with per-env-block episode offsets; MARL reward (its extra sweep runs between the reduction and the district series)."""
    g = golden('g2023_p2')
    spec = g.spec()
    E = 768
    row0 = np.array([0, 24, 100])
    tab = spec.episode_tables(0)
    n_steps = 120
    low, high = spec.action_limits()
    lo, hi = torch.from_numpy(low).cuda()[:, None], torch.from_numpy(high).cuda()[:, None]
    for detail in (True, 'min'):
        kw = dict(kpi=True, detail=detail, reward='MARL', n_steps=n_steps, env_row0=row0)
        fused = StepEngine(tab, E, **kw)
        ref = StepEngine(tab, E, tuning=dict(kpi_passes=1, nw=_kpi_step_waves(E, 3)), **kw)
        fused.trace_kernels(); ref.trace_kernels()
        gen = torch.Generator(device='cuda').manual_seed(3)
        for t in range(30):
            a = (lo + torch.rand((fused.n_act_cols, E), device='cuda', generator=gen) * (hi - lo)).contiguous()
            fused.step(a, t); ref.step(a, t)
        assert fused.last_kernels.startswith('cl_step_full_kpi_kernel<') and 'cl_kpi_kernel' not in fused.last_kernels
        assert ref.last_kernels.endswith('+cl_kpi_kernel')
        assert torch.equal(fused.state, ref.state) and torch.equal(fused.out_env, ref.out_env) and torch.equal(fused.out_bldg[:-1], ref.out_bldg[:-1])
        assert torch.equal(fused.kpi_bldg, ref.kpi_bldg) and torch.equal(fused.kpi_env, ref.kpi_env)
        assert float(fused.kpi_bldg.abs().sum()) > 0


def _selection_map():
    import json
    from pathlib import Path
    f = Path(__file__).resolve().parent / 'golden' / 'kernel_selection_r06.json'
    return json.loads(f.read_text()) if f.exists() else []


@pytest.mark.parametrize('cell', _selection_map(), ids=lambda c: f"{c['kind']}-{c['B']}x{c['E']}")
def test_kernel_selection_map(cell):
    """the launch-geometry rules of `cl_step_f32` (csrc/cl_kernels.hip step_impl) pinned over the whole map they were checked
    on -- B in {3 .. 1024} buildings x E in {4 096 .. 262 144} envs (100 000: not a power of two), battery + PV and thermal districts, the default
    precision model -- not only at the four district sizes they were tuned at.  `tests/golden/kernel_selection_r06.json` is the kernel each cell
    selected in the measuring session (scripts/r06_cliffs.py -> profiles/r06e_cliffs_chain.jsonl, where no forced alternative beat the default by
    more than 10 %): a rule change that moves a cell shows up here and has to come with a new measurement.  Eight cells beyond that map pin the round-6
    rule between the env-major and the building-major kernel (17 x 147 456 ... 2 097 152, 9 / 20 x 1 048 576; profiles/r06_lean_vs_envmajor*.log)."""
    from citylearn_amd.synthetic import tile_district
    base = golden('g2022_all' if cell['kind'] == 'lean' else 'g2020_cz1').spec()
    B, E = cell['B'], cell['E']
    spec = tile_district(base, B, jitter=0.0 if B <= len(base.buildings) else 0.1)
    eng = _StepEngine(spec.episode_tables(0), E, tuning={'finish': 3} if B > 32 else None)
    eng.trace_kernels()
    eng.step(torch.zeros((eng.n_act_cols, E), device='cuda'), 1)
    assert eng.last_kernels == cell['kernel'], (cell, eng.last_kernels)
