"""Host packer of the flexible loads (citylearn_amd/flex.py) on the CPU: table invariants, the reference's schedule-row
convention, refusal of inputs the device tables cannot express, and a table-driven host emulation of the EV SoC rules against
the oracle on synthetic schedules (the device kernel applies the same tables, tests/test_gpu_flex.py)."""
import json
from pathlib import Path

import numpy as np
import pytest

from citylearn_amd import abi
from citylearn_amd.flex import RULE_DRIFT, RULE_KEEP, RULE_ZERO
from citylearn_amd.schema import load_district
from golden_util import golden
from flex_synth import make


def test_tables_are_consistent_with_the_schedules():
    g = golden('g2022_evs')
    spec = g.spec()
    tab = spec.episode_tables(0)
    ft = tab.flex
    T, n_ev = tab.n_steps, len(spec.electric_vehicles)
    assert ft.ev_ts.shape == (T, n_ev, abi.CL_NEVF) and ft.charger_ts.shape[:3] == (T, 7, abi.CL_MAXC)
    assert ft.n_act_cols == sum(len(b.active_actions) for b in spec.buildings) == 26
    # every (row, EV) is advanced by exactly one unit: the charger that holds it, or the EV's own unit
    held = np.zeros((T, n_ev), dtype=int)
    for j in range(len(ft.charger_ids)):
        ev = ft.charger_row(j)[:, abi.CLCT_EV].astype(int)
        for t in np.nonzero(ev >= 0)[0]:
            held[t, ev[t]] += 1
    assert held.max() == 1 and np.array_equal(held, ft.ev_ts[:, :, abi.CLEV_CONNECTED].astype(int))
    # connected EVs never drift; rules are 0, a SoC in [0, 1], or drift; the reset rule keeps the initial SoC or forces one
    step = ft.ev_ts[:, :, abi.CLEV_RULE_STEP]
    assert not np.any((step == RULE_DRIFT) & (held == 1))
    assert np.all((step == RULE_ZERO) | (step == RULE_DRIFT) | ((step >= 0) & (step <= 1)))
    reset = ft.ev_ts[:, :, abi.CLEV_RULE_RESET]
    assert np.all((reset == RULE_KEEP) | ((reset >= 0) & (reset <= 1)))
    assert np.all(ft.ev_ts[-1, :, abi.CLEV_RULE_STEP] == ft.ev_ts[-1, :, abi.CLEV_RULE_LAST])      # no look-ahead past the table
    # empty slots are marked; occupied slots carry the charger's action column
    cols = ft.charger_params.view(np.int32)[:, :, abi.CLC_ACT_COL]
    occupied = ft.charger_ts[0, :, :, abi.CLCT_EV] != -2.0
    assert occupied.sum() == 8 and np.all(cols[occupied] >= 0) and np.all(cols[~occupied] == -1)
    # building flags / plane rows
    flagged = [i for i, row in enumerate(tab.params) if row[abi.CLP_FLAGS] & abi.CLF_FLEX]
    assert flagged == list(ft.flex_bldg) and [int(tab.params.view(np.int32)[i, abi.CLP_FLEX_INDEX]) for i in flagged] == list(range(7))


def test_schedule_rows_follow_the_reference_convention():
    """A later episode still reads the charger schedule from row 0 (the reference never offsets it); an explicit window reads
    its own rows."""
    g = golden('g2022_evs')
    spec = g.spec(episode_time_steps=96)
    first, second = spec.episode_tables(0), spec.episode_tables(1)
    assert (second.start, second.end) == (96, 191)
    assert np.array_equal(first.flex.charger_ts, second.flex.charger_ts) and np.array_equal(first.flex.ev_ts, second.flex.ev_ts)
    window = spec.episode_tables(window=(96, 191))
    whole = spec.episode_tables(window=(spec.simulation_start_time_step, spec.simulation_end_time_step))
    assert np.array_equal(window.flex.charger_ts[:, :, :, abi.CLCT_EV], whole.flex.charger_ts[96:192, :, :, abi.CLCT_EV])
    assert not np.array_equal(window.flex.charger_ts[:, :, :, abi.CLCT_EV], first.flex.charger_ts[:, :, :, abi.CLCT_EV])


def test_inputs_the_tables_cannot_express_are_refused(tmp_path):
    g = golden('g2022_evs')
    schema = json.loads(Path(g.schema_path).read_text())
    schema['root_directory'] = str(Path(g.schema_path).parent)
    b15 = schema['buildings']['Building_15']
    one = b15['chargers']['charger_15_1']
    b15['chargers'].update({f'extra_{k}': dict(one) for k in range(4)})          # 6 chargers on one building
    with pytest.raises(NotImplementedError, match='chargers'):
        load_district(schema).episode_tables(0)
    schema = json.loads(Path(g.schema_path).read_text())
    schema['root_directory'] = str(Path(g.schema_path).parent)
    schema['buildings']['Building_15']['chargers']['charger_15_2']['charger_simulation'] = 'charger_15_1.csv'   # same EV on two chargers
    with pytest.raises(NotImplementedError, match='two chargers'):
        load_district(schema).episode_tables(0)
    schema = json.loads(Path(g.schema_path).read_text())
    schema['root_directory'] = str(Path(g.schema_path).parent)
    schema['buildings']['Building_1']['chargers']['charger_1_1']['attributes']['charge_efficiency_curve'] = [[0.5, 0.9], [0.2, 0.8]]
    with pytest.raises(NotImplementedError, match='increasing'):
        load_district(schema)


@pytest.mark.parametrize('seed', [1, 2, 3])
def test_table_driven_soc_rules_match_the_oracle(seed, tmp_path):
    """Replays the packed begin-of-step rules on the host (what `flex_begin_soc` / `cl_flex_reset_kernel` do with them) with ZERO
    charger actions, where the EV SoC series is a pure function of the tables, against the oracle on synthetic schedules."""
    from oracle.flex_oracle import FlexDistrictOracle
    g = golden('g2022_evs')
    spec = load_district(str(make(Path(g.schema_path).parent, tmp_path / 'synth', seed)))
    for k, ev in enumerate(spec.electric_vehicles):
        ev.battery.initial_soc = 0.15 + 0.1 * k
    tab = spec.episode_tables(0)
    ft, T, n_ev = tab.flex, tab.n_steps, len(spec.electric_vehicles)
    drift = np.random.RandomState(seed).normal(1.0, 0.2, size=(T, n_ev))
    o = FlexDistrictOracle(spec, tab, 1, reward='MARL', drift=drift)
    o.reset()
    init = np.array([ev.battery.initial_soc for ev in spec.electric_vehicles], dtype=np.float32)
    reset = ft.ev_ts[0, :, abi.CLEV_RULE_RESET]
    soc = np.where(reset >= 0, reset, init).astype(np.float32)
    assert np.array_equal(soc, np.array([ev.soc for ev in o.flex[0].evs], dtype=np.float32))
    zero = np.zeros((ft.n_act_cols, 1), dtype=np.float32)
    for t in range(T - 1):
        out = o.step(zero)
        if t > 0:
            rule = ft.ev_ts[t, :, abi.CLEV_RULE_LAST if t + 1 >= T else abi.CLEV_RULE_STEP]
            drifted = np.clip(soc.astype(np.float64) * np.clip(drift[t], 0.6, 1.4), 0.0, 1.0).astype(np.float32)
            soc = np.where(rule >= 0, rule, np.where(rule == RULE_ZERO, np.float32(0.0), drifted)).astype(np.float32)
        np.testing.assert_allclose(soc, out['ev_soc'][:, 0], rtol=0, atol=1e-7, err_msg=f't={t}')
