"""Checkpoint / restore: `state_dict()` / `load_state_dict()` on `StepEngine`,
`VectorCityLearnEnv` and `CityLearnEnv`.  A restored env continues BIT-IDENTICALLY: state planes, output planes, district sums, streaming
KPI accumulators, LSTM rings and hidden state, EV / washing-machine state, host-side histories.  The reference's counterpart is pickling
the env object (citylearn/__main__.py:291-299).  GPU only."""
import io

import numpy as np
import pytest
import torch

from golden_util import golden
from citylearn_amd import abi
from citylearn_amd.engine import StepEngine

pytestmark = pytest.mark.gpu


def _roundtrip(sd):
    """through torch.save / torch.load: what a user would do"""
    buf = io.BytesIO()
    torch.save(sd, buf)
    buf.seek(0)
    return torch.load(buf, weights_only=False)


def _engine_tensors(e):
    out = [e._state_store, e._out_store[:abi.CLO_RESERVED], e.out_env]
    if e.kpi_bldg is not None:
        out += [e.kpi_bldg, e.kpi_env]
    if e.flex is not None:
        out += [e.ev_state, e.wm_state, e.flex_out]
    return out


@pytest.mark.parametrize('name,kw', [('g2022_all', dict(kpi=True)), ('g2022_all', dict(f64_maps=False)), ('g2020_cz1', dict(kpi=True, detail=True)),
                                     ('g2023_p2', dict(reward='MARL')), ('g2022_all', dict(f64_maps=True))])
def test_engine_restored_continues_bit_identically(name, kw):
    g = golden(name)
    spec = g.spec()
    tab = spec.episode_tables(0)
    E = 260
    low, high = spec.action_limits()
    lo, hi = torch.from_numpy(low).cuda()[:, None], torch.from_numpy(high).cuda()[:, None]
    gen = torch.Generator(device='cuda').manual_seed(7)
    acts = [(lo + torch.rand((len(low), E), device='cuda', generator=gen) * (hi - lo)).contiguous() for _ in range(30)]
    a = StepEngine(tab, E, **kw)
    for t in range(20):
        a.step(acts[t], t)
    sd = _roundtrip(a.state_dict())
    b = StepEngine(tab, E, **kw)
    b.step(acts[5], 0)                                        # (some other history in the target's buffers)
    b.load_state_dict(sd)
    assert b.t == 20
    for x, y in zip(_engine_tensors(a), _engine_tensors(b)):
        assert torch.equal(x, y)
    for t in range(20, 30):
        a.step(acts[t], t); b.step(acts[t], t)
        for x, y in zip(_engine_tensors(a), _engine_tensors(b)):
            assert torch.equal(x, y), t
    # a checkpoint of another batch size / another district is refused
    with pytest.raises(ValueError):
        StepEngine(tab, E + 4, **kw).load_state_dict(sd)


def test_engine_checkpoint_under_the_deferred_finish():
    """Building-chunked district with `tuning={'finish': 3}`: the checkpoint folds the pending district sums first and carries the scratch rows
    and marker words, so the next (deferring) launch of the restored engine folds the same partial sums."""
    from citylearn_amd.synthetic import tile_district
    spec = tile_district(golden('g2022_all').spec(), 100)
    tab = spec.episode_tables(0)
    E = 512
    gen = torch.Generator(device='cuda').manual_seed(1)
    acts = [torch.rand((100, E), device='cuda', generator=gen) * 2 - 1 for _ in range(12)]
    a = StepEngine(tab, E, f64_maps=False, tuning=dict(finish=3))
    for t in range(7):
        a.step(acts[t], t)
    sd = a.state_dict()
    b = StepEngine(tab, E, f64_maps=False, tuning=dict(finish=3))
    b.load_state_dict(sd)
    for t in range(7, 12):
        a.step(acts[t], t); b.step(acts[t], t)
        assert torch.equal(a.state, b.state) and torch.equal(a.out_bldg[:2], b.out_bldg[:2]) and torch.equal(a.out_env, b.out_env), t


@pytest.mark.parametrize('name,kw', [('g2023_p2', dict(kpi=True)), ('g2022_evs', dict()), ('g2022_all', dict(observations='tensor', normalize_observations=True)),
                                     ('s_baeda', dict())])
def test_vector_env_restored_continues_bit_identically(name, kw):
    """LSTM stage + ComfortReward + comfort KPIs (2023), EV chargers with the on-device drift stream (2022 + EVs), the observation tensor,
    the generic-shape LSTM (baeda): a fresh env restored from the checkpoint returns the same observations and rewards and ends with the same KPIs."""
    from citylearn_amd.vector_env import VectorCityLearnEnv
    g = golden(name)
    E = 64
    a = VectorCityLearnEnv(g.schema_path, n_envs=E, **kw)
    gen = torch.Generator(device='cuda').manual_seed(11)
    K1, K2 = 30, 15
    acts = [a.sample_actions(gen) for _ in range(K1 + K2)]
    for t in range(K1):
        a.step(acts[t])
    sd = _roundtrip(a.state_dict())
    b = VectorCityLearnEnv(g.schema_path, n_envs=E, **kw)
    b.step(acts[3])
    b.load_state_dict(sd)
    assert b.time_step == a.time_step == K1

    def same(x, y):
        if isinstance(x, dict):
            return all(same(x[k], y[k]) for k in x)
        return torch.equal(x, y)
    for t in range(K1, K1 + K2):
        oa, ra, *_ = a.step(acts[t])
        ob, rb, *_ = b.step(acts[t])
        assert same(oa, ob) and torch.equal(ra, rb), t
    if kw.get('kpi'):
        (ba, da), (bb, db) = a.evaluate(), b.evaluate()
        for k in ba:
            assert torch.equal(torch.nan_to_num(ba[k]), torch.nan_to_num(bb[k])), k
        for k in da:
            assert torch.equal(torch.nan_to_num(da[k]), torch.nan_to_num(db[k])), k


def test_vector_env_checkpoint_rebuilds_the_saved_episode():
    """A checkpoint taken in episode 2 of a split schedule restores into an env standing in episode 0: the saved episode's window is rebuilt first."""
    from citylearn_amd.vector_env import VectorCityLearnEnv
    g = golden('g2022_all')
    kw = dict(episode_time_steps=48, rolling_episode_split=True)
    a = VectorCityLearnEnv(g.schema_path, n_envs=32, **kw)
    a.reset(); a.reset()
    gen = torch.Generator(device='cuda').manual_seed(2)
    acts = [a.sample_actions(gen) for _ in range(20)]
    for t in range(10):
        a.step(acts[t])
    sd = a.state_dict()
    b = VectorCityLearnEnv(g.schema_path, n_envs=32, **kw)
    assert (b.tables.start, b.tables.end) != (a.tables.start, a.tables.end)
    b.load_state_dict(sd)
    assert (b.tables.start, b.tables.end) == (a.tables.start, a.tables.end) and b.time_step == 10
    for t in range(10, 20):
        oa, ra, *_ = a.step(acts[t]); ob, rb, *_ = b.step(acts[t])
        assert torch.equal(ra, rb) and all(torch.equal(oa[k], ob[k]) for k in oa), t


@pytest.mark.parametrize('name', ['g2022_all', 'g2023_p2'])
def test_single_district_env_checkpoint(name):
    """`CityLearnEnv` (lists in, lists out): observations, rewards and the `evaluate()` frame of a restored env equal the original's."""
    from citylearn_amd.citylearn import CityLearnEnv
    g = golden(name)
    a = CityLearnEnv(g.schema_path)
    a.reset()

    def act(env, t):
        v = [float(x) for x in g.ref['actions'][t]]
        if env.central_agent:
            return [v]
        out, p = [], 0
        for names in env.action_names:
            out.append(v[p:p + len(names)]); p += len(names)
        return out
    for t in range(40):
        a.step(act(a, t))
    sd = _roundtrip(a.state_dict())
    b = CityLearnEnv(g.schema_path)
    b.reset()
    b.step(act(b, 0))
    b.load_state_dict(sd)
    assert b.time_step == 40 and b.observations == a.observations
    for t in range(40, 60):
        ra, rb = a.step(act(a, t)), b.step(act(b, t))
        assert ra[0] == rb[0] and ra[1] == rb[1] and ra[2] == rb[2], t
    fa, fb = a.evaluate(), b.evaluate()
    assert list(fa['cost_function']) == list(fb['cost_function'])
    np.testing.assert_array_equal(np.nan_to_num(fa['value'].to_numpy(dtype=float)), np.nan_to_num(fb['value'].to_numpy(dtype=float)))
