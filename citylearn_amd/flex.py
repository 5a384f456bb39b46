"""Host packer for the flexible loads of a district: EV chargers, electric vehicles, washing machines.

Everything the reference decides from the charger / washing-machine schedules alone -- which EV a charger holds, when
an arrival SoC is forced, when an unconnected EV drifts, when a washing-machine window opens -- is evaluated here once
per table row and handed to `cl_flex_kernel` (csrc/cl_flex.h) as small tables; the device keeps what depends on the
actions (EV battery state, washing-machine `initiated` flags).  Layouts: include/citylearn_amd.h (`cl_flex`).

Reference paths: citylearn.py:1353-1474 (association / unconnected-EV rules), electric_vehicle_charger.py:297-334,
energy_model.py:1289-1330 (washing machine), building.py:1221-1296 (charger observations).

Schedule rows.  The reference never gives charger / washing-machine simulations an episode offset (building.py:2601-2618
resets four data sets; these are not among them), so episode step t always reads schedule row t counted from
`simulation_start_time_step`.  ``aligned=False`` reproduces that; ``aligned=True`` reads the rows of the episode window
like every other series (used with per-env-block episode offsets, where the tables span the simulation period).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import abi

RULE_ZERO, RULE_DRIFT, RULE_KEEP = -1.0, -2.0, -3.0          # CLEV_ZERO / CLEV_DRIFT / CLEV_KEEP
DEFAULT_EV_REWARD_WEIGHTS = {      # reward_function.py:399-407
    'no_car_charging': -5.0, 'battery_limits': -2.0, 'soc_impossible': -10.0, 'soc_under': -5.0, 'close_soc': 10.0,
    'self_ev_consumption': 5.0, 'extra_self_production': 5.0,
}


@dataclass
class FlexTables:
    ev_names: List[str]
    charger_ids: List[Tuple[int, str]]          # (building index, charger id), building-major
    wm_names: List[Tuple[int, str]]
    ev_params: np.ndarray                       # uint32 [n_ev, CL_NP]
    ev_ts: np.ndarray                           # float32 [R, n_ev, CL_NEVF]
    charger_params: np.ndarray                  # uint32 [n_flex_bldg, CL_MAXC, CL_NCP]   (slot tables, see the header)
    charger_ts: np.ndarray                      # float32 [R, n_flex_bldg, CL_MAXC, CL_NCF]
    wm_params: np.ndarray                       # uint32 [n_flex_bldg, CL_MAXW, CL_NWP]
    wm_ts: np.ndarray                           # float32 [R, n_flex_bldg, CL_MAXW, CL_NWF]
    flex_bldg: np.ndarray                       # int32 [n_flex_bldg]: building index of every row of the flexible-load planes
    cons_params: Optional[np.ndarray]           # uint32 [n_flex_bldg, CL_NCC] charging constraints, None when no building has any
    charger_slot: np.ndarray                    # int32 [n_charger]: flex row * CL_MAXC + slot of charger j (index into charger_out / the tables)
    wm_slot: np.ndarray                         # int32 [n_wm]
    n_act_cols: int
    observations: Dict[str, np.ndarray] = field(default_factory=dict)   # name -> [R] values after a step; see `reset_observations`
    reset_observations: Dict[str, np.ndarray] = field(default_factory=dict)   # name -> [R] values when the episode starts on that row

    @property
    def n_rows(self) -> int:
        return self.charger_ts.shape[0]

    def charger_row(self, j: int) -> np.ndarray:
        """[R, CL_NCF] schedule rows of charger j."""
        fb, slot = divmod(int(self.charger_slot[j]), abi.CL_MAXC)
        return self.charger_ts[:, fb, slot]


def _battery_block(battery, r: float) -> np.ndarray:
    """The CLP_L_* words of `schema.episode_tables` for one EV battery."""
    row = np.zeros(abi.CL_NP, dtype=np.uint32)
    pf = row.view(np.float32)
    cap, powr = float(battery.capacity), float(battery.nominal_power)
    pf[abi.CLP_L_TSR] = r
    pf[abi.CLP_L_PDT] = powr
    pf[abi.CLP_L_POW] = powr
    pf[abi.CLP_L_CAP] = cap
    pf[abi.CLP_L_CAPL] = cap * (1.0 - battery.loss_coefficient * r)
    pf[abi.CLP_L_INV_CAP] = 1.0 / max(cap, 1e-6)
    pf[abi.CLP_L_INV_POW] = 1.0 / max(powr, 1e-6)
    pf[abi.CLP_L_OMD] = 1.0 - battery.depth_of_discharge
    pf[abi.CLP_L_DEGK] = battery.capacity_loss_coefficient * cap * r / 2.0
    cx, cy = np.asarray(battery.capacity_power_curve, dtype=float)
    pf[abi.CLP_L_CPC_X1] = cx[1]
    for k, (sa, sb) in enumerate(((abi.CLP_L_CPC_A0, abi.CLP_L_CPC_B0), (abi.CLP_L_CPC_A1, abi.CLP_L_CPC_B1))):
        slope = (cy[k + 1] - cy[k]) / (cx[k + 1] - cx[k])
        pf[sa] = powr * (cy[k] - slope * cx[k])
        pf[sb] = powr * slope
    ex, ey = np.asarray(battery.power_efficiency_curve, dtype=float)
    pf[abi.CLP_L_PEC_X1:abi.CLP_L_PEC_X1 + 3] = ex[1:4]
    for k in range(4):
        slope = (ey[k + 1] - ey[k]) / (ex[k + 1] - ex[k])
        pf[abi.CLP_L_PEC_A0 + 2 * k] = ey[k] - slope * ex[k]
        pf[abi.CLP_L_PEC_B0 + 2 * k] = slope
    pf[abi.CLP_L_SOC0] = battery.initial_soc
    pf[abi.CLP_L_EFF0] = battery.efficiency
    return row


def _arrival_soc(sim, step: int, prev_state, prev_id, ev_id) -> Optional[float]:
    """`_resolve_arrival_soc` (citylearn.py:1356-1376)."""
    idx = step - 1 if (prev_state == 2 and step > 0 and isinstance(prev_id, str) and prev_id == ev_id) else step
    arr = sim['electric_vehicle_estimated_soc_arrival']
    v = arr[idx] if 0 <= idx < len(arr) else np.nan
    if not np.isnan(v) and 0.0 <= v <= 1.0:
        return float(v)
    cur = sim['current_soc']
    j = min(step, len(cur) - 1)
    if j >= 0 and not np.isnan(cur[j]) and 0.0 <= cur[j] <= 1.0:
        return float(cur[j])
    return None


def pack_flex(spec, start: int, n_rows: int, aligned: bool = False) -> Optional[FlexTables]:
    """Tables for schedule rows of one episode window [start, start + n_rows) of the data files.

    ``aligned=False`` (the reference's behaviour): rows 0 .. n_rows-1 of the charger / washing-machine schedules,
    whatever the window's start."""
    if not spec.has_flexible_loads:
        return None
    first = (start - spec.simulation_start_time_step) if aligned else 0
    R = int(n_rows)
    names = [ev.name for ev in spec.electric_vehicles]
    chargers = [(i, c) for i, b in enumerate(spec.buildings) for c in b.chargers]
    wms = [(i, w) for i, b in enumerate(spec.buildings) for w in b.washing_machines]
    n_ev, n_c, n_w = len(names), len(chargers), len(wms)
    r = float(spec.buildings[0].time_step_ratio)
    for _, c in chargers:
        if first + R > len(c.series['electric_vehicle_charger_state']):
            raise ValueError(f'charger {c.charger_id}: schedule shorter than the episode window')

    # action columns: building-major over each building's active actions (citylearn.py:1069-1079)
    col_of: Dict[Tuple[int, str], int] = {}
    col = 0
    for i, b in enumerate(spec.buildings):
        for k in b.active_actions:
            col_of[(i, k)] = col
            col += 1

    ev_params = np.stack([_battery_block(ev.battery, r) for ev in spec.electric_vehicles]) if n_ev else np.zeros((0, abi.CL_NP), np.uint32)
    ev_ts = np.zeros((R, n_ev, abi.CL_NEVF), dtype=np.float32)
    ev_ts[:, :, abi.CLEV_RULE_STEP] = RULE_ZERO
    ev_ts[:, :, abi.CLEV_RULE_LAST] = RULE_ZERO
    ev_ts[:, :, abi.CLEV_RULE_RESET] = RULE_KEEP
    flex_b = [i for i, b in enumerate(spec.buildings) if b.chargers or b.washing_machines]
    n_fb = len(flex_b)
    for i in flex_b:
        b = spec.buildings[i]
        if len(b.chargers) > abi.CL_MAXC or len(b.washing_machines) > abi.CL_MAXW:
            raise NotImplementedError(f'{b.name}: at most {abi.CL_MAXC} chargers and {abi.CL_MAXW} washing machines per building '
                                      f'(got {len(b.chargers)} / {len(b.washing_machines)})')
    charger_slot = np.array([flex_b.index(i) * abi.CL_MAXC + [id(x) for x in spec.buildings[i].chargers].index(id(c))
                             for i, c in chargers], dtype=np.int32)
    wm_slot = np.array([flex_b.index(i) * abi.CL_MAXW + [id(x) for x in spec.buildings[i].washing_machines].index(id(w))
                        for i, w in wms], dtype=np.int32)
    charger_params = np.zeros((n_fb * abi.CL_MAXC, abi.CL_NCP), dtype=np.uint32)
    charger_params.view(np.int32)[:, abi.CLC_ACT_COL] = -1
    charger_ts = np.zeros((R, n_fb * abi.CL_MAXC, abi.CL_NCF), dtype=np.float32)
    charger_ts[:, :, abi.CLCT_EV] = -2.0                       # CLCT_EMPTY
    charger_ts[:, charger_slot, abi.CLCT_EV] = -1.0
    cpf, cpi = charger_params.view(np.float32), charger_params.view(np.int32)
    for jj, (i, c) in enumerate(chargers):
        j = int(charger_slot[jj])
        cpi[j, abi.CLC_ACT_COL] = col_of.get((i, c.action_name), -1)
        cpf[j, abi.CLC_MAX_CHARGE], cpf[j, abi.CLC_MIN_CHARGE] = c.max_charging_power, c.min_charging_power
        cpf[j, abi.CLC_MAX_DISCHARGE], cpf[j, abi.CLC_MIN_DISCHARGE] = c.max_discharging_power, c.min_discharging_power
        cpf[j, abi.CLC_EFF], cpf[j, abi.CLC_INV_EFF] = c.efficiency, 1.0 / c.efficiency
        cpf[j, abi.CLC_DT_HOURS] = spec.seconds_per_time_step / 3600.0
        for curve, n_slot, x_slot, y_slot in ((c.charge_efficiency_curve, abi.CLC_CURVE_CHARGE_N, abi.CLC_CURVE_CHARGE_X, abi.CLC_CURVE_CHARGE_Y),
                                              (c.discharge_efficiency_curve, abi.CLC_CURVE_DISCHARGE_N, abi.CLC_CURVE_DISCHARGE_X, abi.CLC_CURVE_DISCHARGE_Y)):
            if curve is not None:
                k = curve.shape[1]
                cpi[j, n_slot] = k
                cpf[j, x_slot:x_slot + k] = curve[0]
                cpf[j, y_slot:y_slot + k] = curve[1]
        w = slice(first, first + R)
        charger_ts[:, j, abi.CLCT_REQUIRED_SOC] = c.series['electric_vehicle_required_soc_departure'][w]
        charger_ts[:, j, abi.CLCT_DEPARTURE] = c.series['electric_vehicle_departure_time'][w]

    sims = [{k: v[first:first + R] for k, v in c.series.items()} for _, c in chargers]
    holder = np.full((R, n_ev), -1, dtype=np.int64)
    for rho in range(R):
        for k, name in enumerate(names):
            # -- CityLearnEnv.simulate_unconnected_ev_soc entered from row rho - 1 (citylearn.py:1416-1474) --
            step_rule = RULE_ZERO
            if rho + 1 < R:
                found = False
                for sim in sims:
                    ids, st = sim['electric_vehicle_id'], sim['electric_vehicle_charger_state']
                    cur_id, nxt_id, cur_st, nxt_st = ids[rho], ids[rho + 1], st[rho], st[rho + 1]
                    if isinstance(cur_id, str) and cur_id == name and cur_st == 1:
                        found = True
                        break
                    connecting = isinstance(nxt_id, str) and nxt_id == name and nxt_st == 1 and cur_st != 1
                    incoming = isinstance(cur_id, str) and cur_id == name and cur_st == 2
                    if connecting:
                        found = True
                        arr = sim['electric_vehicle_estimated_soc_arrival']
                        soc = arr[rho] if incoming else arr[rho + 1]
                        if 0 <= soc <= 1:
                            step_rule = float(soc)
                        break
                if not found:
                    step_rule = RULE_DRIFT
            last_rule, reset_rule = RULE_ZERO, RULE_KEEP
            # -- CityLearnEnv.associate_chargers_to_electric_vehicles on row rho (citylearn.py:1353-1414) --
            for j, sim in enumerate(sims):
                state = sim['electric_vehicle_charger_state'][rho]
                ev_id = sim['electric_vehicle_id'][rho]
                if np.isnan(state) or state != 1 or not isinstance(ev_id, str) or ev_id != name:
                    continue
                if holder[rho, k] >= 0:
                    raise NotImplementedError(f'{name} is plugged into two chargers on schedule row {first + rho}')
                holder[rho, k] = j
                charger_ts[rho, charger_slot[j], abi.CLCT_EV] = k
                prev_state = sim['electric_vehicle_charger_state'][rho - 1] if rho > 0 else np.nan
                prev_id = sim['electric_vehicle_id'][rho - 1] if rho > 0 else None
                if prev_state != 1 or not isinstance(prev_id, str) or prev_id != ev_id:
                    soc = _arrival_soc(sim, rho, prev_state, prev_id, ev_id)
                    if soc is not None:
                        step_rule = last_rule = soc
                soc = _arrival_soc(sim, rho, np.nan, None, ev_id)      # an episode starting here: always a new connection
                if soc is not None:
                    reset_rule = soc
            if rho + 1 >= R:
                step_rule = last_rule
            ev_ts[rho, k] = (step_rule, last_rule, reset_rule, 1.0 if holder[rho, k] >= 0 else 0.0)
            if holder[rho, k] >= 0:            # the charger row carries its EV's rules: no dependent table read on the device
                charger_ts[rho, charger_slot[holder[rho, k]], abi.CLCT_RULE_STEP] = step_rule
                charger_ts[rho, charger_slot[holder[rho, k]], abi.CLCT_RULE_LAST] = last_rule

    wm_params = np.zeros((n_fb * abi.CL_MAXW, abi.CL_NWP), dtype=np.uint32)
    wm_params.view(np.int32)[:, 0] = -1
    wm_ts = np.zeros((R, n_fb * abi.CL_MAXW, abi.CL_NWF), dtype=np.float32)
    wm_ts[:, :, abi.CLWT_OPEN] = -1.0                          # CLWT_EMPTY
    for jj, (i, w) in enumerate(wms):
        j = int(wm_slot[jj])
        wm_params.view(np.int32)[j, 0] = col_of.get((i, w.name), -1)
        s = w.series['wm_start_time_step'][first:first + R]
        e = w.series['wm_end_time_step'][first:first + R]
        profiles = w.series['load_profile'][first:first + R]
        for rho in range(R):
            step = rho if not aligned else first + rho          # the reference compares with the episode step (energy_model.py:1320)
            has = len(profiles[rho]) > 0
            wm_ts[rho, j, abi.CLWT_OPEN] = float(has and s[rho] != -1 and e[rho] != -1 and s[rho] <= step <= e[rho])
            wm_ts[rho, j, abi.CLWT_NEW_WINDOW] = float(rho > 0 and (s[rho - 1] != s[rho] or e[rho - 1] != e[rho]))
            wm_ts[rho, j, abi.CLWT_LOAD] = float(np.float32(sum(np.float32(v) for o, v in enumerate(profiles[rho]) if rho + o < R))) if has else 0.0

    cons_params = None
    if any(spec.buildings[i].charging_constraints is not None for i in flex_b):
        cons_params = np.zeros((n_fb, abi.CL_NCC), dtype=np.uint32)
        cf = cons_params.view(np.float32)
        cf[:, abi.CLCC_BUILDING_LIMIT:abi.CLCC_PHASE_LIMIT0 + abi.CL_MAXPH] = -1.0
        for fb, i in enumerate(flex_b):
            b = spec.buildings[i]
            cc = b.charging_constraints
            if cc is None:
                continue
            ids = [c.charger_id for c in b.chargers]
            cons_params[fb, abi.CLCC_FLAGS] = 1
            cf[fb, abi.CLCC_BUILDING_LIMIT] = -1.0 if cc.building_limit_kw is None else cc.building_limit_kw
            for p_, ph in enumerate(cc.phases):
                cf[fb, abi.CLCC_PHASE_LIMIT0 + p_] = -1.0 if ph['limit_kw'] is None else ph['limit_kw']
                cons_params[fb, abi.CLCC_PHASE_MASK0 + p_] = sum(1 << ids.index(cid) for cid in ph['chargers'] if cid in ids)

    out = FlexTables(ev_names=names, charger_ids=[(i, c.charger_id) for i, c in chargers], wm_names=[(i, w.name) for i, w in wms],
                     ev_params=ev_params, ev_ts=ev_ts,
                     charger_params=charger_params.reshape(n_fb, abi.CL_MAXC, abi.CL_NCP),
                     charger_ts=charger_ts.reshape(R, n_fb, abi.CL_MAXC, abi.CL_NCF),
                     wm_params=wm_params.reshape(n_fb, abi.CL_MAXW, abi.CL_NWP), wm_ts=wm_ts.reshape(R, n_fb, abi.CL_MAXW, abi.CL_NWF),
                     flex_bldg=np.array(flex_b, dtype=np.int32), cons_params=cons_params, charger_slot=charger_slot, wm_slot=wm_slot, n_act_cols=col)
    _pack_observations(out, spec, sims, chargers, wms, first, R)
    return out


def _pack_observations(tab: FlexTables, spec, sims, chargers, wms, first: int, R: int) -> None:
    """`Building.update_ev_charger_observations` / `update_washing_machine_observations` (building.py:1221-1334): all of
    them are functions of the schedule row.  The connected EV's SoC is the entry `soc[row]` BEFORE any action on that
    row: an arrival SoC on the first row of a connection, 0 afterwards (nothing carries the charged value forward)."""
    for variant, rule_col, dst in (('step', abi.CLEV_RULE_STEP, tab.observations), ('reset', abi.CLEV_RULE_RESET, tab.reset_observations)):
        for j, (i, c) in enumerate(chargers):
            sim, cid = sims[j], c.charger_id
            state = sim['electric_vehicle_charger_state']
            ev = tab.charger_row(j)[:, abi.CLCT_EV].astype(int)
            connected = (state == 1) & (ev >= 0)
            incoming_ids = np.array([isinstance(v, str) and v in tab.ev_names for v in sim['electric_vehicle_id']])
            incoming = (state == 2) & incoming_ids
            soc = np.full(R, -0.1)
            for rho in np.nonzero(connected)[0]:
                rule = float(tab.ev_ts[rho, ev[rho], rule_col])
                if rule == RULE_KEEP:
                    rule = spec.electric_vehicles[ev[rho]].battery.initial_soc
                soc[rho] = np.float32(max(rule, 0.0)) if rule != RULE_DRIFT else np.nan
            dst[f'electric_vehicle_charger_{cid}_connected_state'] = connected.astype(float)
            dst[f'connected_electric_vehicle_at_charger_{cid}_departure_time'] = np.where(connected, sim['electric_vehicle_departure_time'].astype(float), -1.0)
            dst[f'connected_electric_vehicle_at_charger_{cid}_required_soc_departure'] = np.where(connected, sim['electric_vehicle_required_soc_departure'], -0.1)
            dst[f'connected_electric_vehicle_at_charger_{cid}_soc'] = soc
            dst[f'connected_electric_vehicle_at_charger_{cid}_battery_capacity'] = np.where(connected, sim['electric_vehicle_battery_capacity_kwh'], -1.0)
            dst[f'electric_vehicle_charger_{cid}_incoming_state'] = incoming.astype(float)
            dst[f'incoming_electric_vehicle_at_charger_{cid}_estimated_arrival_time'] = np.where(incoming, sim['electric_vehicle_estimated_arrival_time'].astype(float), -1.0)
            dst[f'incoming_electric_vehicle_at_charger_{cid}_estimated_soc_arrival'] = np.where(incoming, sim['electric_vehicle_estimated_soc_arrival'], -0.1)
        for j, (i, w) in enumerate(wms):
            dst[f'{w.name}_start_time_step'] = w.series['wm_start_time_step'][first:first + R].astype(float)
            dst[f'{w.name}_end_time_step'] = w.series['wm_end_time_step'][first:first + R].astype(float)


def reward_weights(weights=None, penalty_coefficient: float = 1.0) -> np.ndarray:
    """`cl_flex.weights` from the `weights` mapping of Electric_Vehicles_Reward_Function (reward_function.py:396-407)."""
    w = dict(DEFAULT_EV_REWARD_WEIGHTS if not weights else weights)
    out = np.zeros(8, dtype=np.float32)
    out[abi.CLEW_BATTERY_LIMITS] = w['battery_limits']
    out[abi.CLEW_SOC_IMPOSSIBLE] = w['soc_impossible']
    out[abi.CLEW_SOC_UNDER] = w['soc_under']
    out[abi.CLEW_CLOSE_SOC] = w['close_soc']
    out[abi.CLEW_SELF_EV_CONSUMPTION] = w['self_ev_consumption']
    out[abi.CLEW_EXTRA_SELF_PRODUCTION] = w['extra_self_production']
    out[abi.CLEW_PENALTY_COEFFICIENT] = 1.0 if penalty_coefficient is None else penalty_coefficient   # reward_function.py:49-58
    return out
