"""CPU oracle for EV chargers and washing machines (TEST INFRASTRUCTURE, NOT PRODUCT CODE).

Scalar restatement of the reference's electric-vehicle / charger / washing-machine step on top of
`oracle.DistrictOracle`.  Pinned against a trajectory of the reference itself on its
``citylearn_challenge_2022_phase_all_plus_evs`` dataset (`tests/golden/g2022_evs`, generator
`oracle/ref_harness/gen_golden.py`).  Paths below are relative to /root/reference/citylearn/.

What the reference does per environment step t (citylearn.py:978-1056):

1. `Building.apply_actions` runs the charger and washing-machine actions after every other device
   (building.py:1581-1605): `Charger.update_connected_electric_vehicle_soc` (electric_vehicle_charger.py:297-334) and
   `WashingMachine.start_cycle` (energy_model.py:1314-1330).
2. `Building.update_variables` adds the chargers' and washing machines' consumption to the net (building.py:2657-2693).
3. the reward sees the charger dictionaries of `Building._get_observations_data` (building.py:1340-1389).
4. `CityLearnEnv.next_time_step` (citylearn.py:1325-1351): devices advance, then `simulate_unconnected_ev_soc`
   (1416-1474) and `associate_chargers_to_electric_vehicles` (1353-1414) write the NEW step's SoC entry of some EVs.

SoC series are zero-initialised float32 arrays and nothing carries a value forward: an entry is only written by
`Battery.charge` (non-zero charger action on a connected EV) or `force_set_soc` (arrival / drift rules).
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np

from .oracle import DistrictOracle, F32, _Battery, reward_values

EV_REWARD_WEIGHTS = {   # reward_function.py:399-407
    'no_car_charging': -5.0, 'battery_limits': -2.0, 'soc_impossible': -10.0, 'soc_under': -5.0, 'close_soc': 10.0,
    'self_ev_consumption': 5.0, 'extra_self_production': 5.0,
}


class _EVBattery(_Battery):
    """`ElectricVehicle.battery`: a `Battery` whose SoC entries can also be written by `force_set_soc`
    (energy_model.py:1111-1128)."""

    def __init__(self, spec, r):
        _Battery.__init__(self, spec, r)
        self.t = 0

    def begin_step(self, t: int):
        _Battery.begin_step(self, t)
        self.t = t

    def force_set_soc(self, soc):
        if soc < 0 or soc > 1:
            raise AttributeError('Soc must be between 0 and 1. Check your dataset')
        self.soc = F32(soc)
        if self.t == 0:
            self.prev_soc = self.soc      # at t = 0 `energy_init` / `soc_init` read soc[0] itself (energy_model.py:661-666, 1047)


class _Charger:
    def __init__(self, cspec, n_steps: int, dt_hours: float):
        self.spec = cspec
        self.sim = cspec.series
        self.dt_hours = dt_hours      # algorithm_action_based_time_step_hours_ratio (electric_vehicle_charger.py:62-63)
        self.connected: Optional[int] = None
        self.incoming: Optional[int] = None
        self.consumption = F32(0.0)
        self.energy_kwh = F32(0.0)

    def begin_step(self):
        # Charger.next_time_step (electric_vehicle_charger.py:336-340) + fresh zero slots of the float32 series
        self.connected = None
        self.incoming = None
        self.consumption = F32(0.0)
        self.energy_kwh = F32(0.0)

    def update(self, action: float, evs: List[_EVBattery]):
        """electric_vehicle_charger.py:297-334 (no efficiency curves: `get_efficiency` returns `efficiency`)."""
        c = self.spec
        if action != 0:
            charging = action > 0
            curve = c.charge_efficiency_curve if charging else c.discharge_efficiency_curve
            eff = c.efficiency if curve is None else np.interp(abs(action), curve[0], curve[1])     # get_efficiency (264-295)
            if charging:
                power = action * c.max_charging_power
                energy = power * self.dt_hours
                energy = max(min(energy, c.max_charging_power), c.min_charging_power)
                energy_kwh = energy * eff
            else:
                power = action * c.max_discharging_power
                energy = power * self.dt_hours
                energy = max(min(energy, -c.min_discharging_power), -c.max_discharging_power)
                energy_kwh = energy / eff
            self.energy_kwh = F32(energy)
            if self.connected is not None:
                battery = evs[self.connected]
                battery.charge(energy_kwh)
                eb = battery.eb
                self.consumption = F32(eb / eff if eb >= 0 else eb * eff)
            else:
                self.consumption = F32(0.0)
        else:
            self.consumption = F32(0.0)
            self.energy_kwh = F32(0.0)


class _WashingMachine:
    def __init__(self, wspec, n_steps: int):
        self.spec = wspec
        self.sim = wspec.series
        self.n_steps = n_steps
        self.initiated = False
        self.consumption = F32(0.0)

    def reset(self):
        self.initiated = False
        self.consumption = F32(0.0)

    def begin_step(self, t: int):
        """WashingMachine.next_time_step (energy_model.py:1289-1312): a new window clears `initiated`."""
        self.consumption = F32(0.0)
        if t > 0:
            s, e = self.sim['wm_start_time_step'], self.sim['wm_end_time_step']
            if (s[t - 1] != s[t] or e[t - 1] != e[t]) and self.initiated:
                self.initiated = False

    def start_cycle(self, action: float, t: int):
        """energy_model.py:1314-1330: the whole load profile is booked on the CURRENT step's slot."""
        s, e = self.sim['wm_start_time_step'][t], self.sim['wm_end_time_step'][t]
        if not self.initiated and action > 0 and s != -1 and e != -1 and s <= t <= e:
            profile = self.sim['load_profile'][t]
            if len(profile) == 0:
                return
            self.initiated = True
            for offset, load in enumerate(profile):
                if t + offset < self.n_steps:
                    self.consumption = F32(self.consumption + load)


def apply_charging_constraints(bspec, actions: Dict[str, float], dt_hours: float):
    """`Building._apply_charging_constraints_to_actions` (building.py:901-989).  `actions`: charger id -> action.
    Returns (actions, violation_kwh, headroom) with headroom = {'building': kW or None, phase name: kW or None}."""
    cc = bspec.charging_constraints
    default = {'building': None if cc.building_limit_kw is None else float(cc.building_limit_kw),
               **{ph['name']: None if ph['limit_kw'] is None else float(ph['limit_kw']) for ph in cc.phases}}
    if not actions:
        return actions, 0.0, default
    lookup = {c.charger_id: c for c in bspec.chargers}
    requests, scales = {}, {}
    for cid, a in actions.items():
        if a is None or a <= 0.0 or cid not in lookup:
            continue
        max_power = lookup[cid].max_charging_power or 0.0
        if max_power <= 0.0:
            continue
        requests[cid] = a * max_power
        scales[cid] = 1.0
    violation_kw = 0.0
    if not requests:
        return actions, 0.0, default
    total = sum(requests.values())
    limit = cc.building_limit_kw
    if limit is not None and limit >= 0.0 and total > limit:
        scale = 0.0 if limit == 0 else limit / total
        for cid in scales:
            scales[cid] *= scale
        violation_kw += total - limit
    for ph in cc.phases:
        limit = ph['limit_kw']
        if limit is None or limit < 0.0:
            continue
        phase_sum = sum(requests.get(cid, 0.0) * scales.get(cid, 1.0) for cid in ph['chargers'] if cid in requests)
        if phase_sum > limit:
            f = 0.0 if limit == 0 else limit / phase_sum
            for cid in ph['chargers']:
                if cid in scales:
                    scales[cid] *= f
            violation_kw += phase_sum - limit
    scaled = {cid: requests[cid] * scales.get(cid, 1.0) for cid in requests}
    actions = dict(actions)
    for cid, a in list(actions.items()):
        if a is None or a <= 0.0 or cid not in lookup:
            continue
        max_power = lookup[cid].max_charging_power or 0.0
        if max_power <= 0.0:
            actions[cid] = 0.0
            continue
        actions[cid] = max(0.0, min(a, scaled.get(cid, 0.0) / max_power))
    headroom = dict(default)
    if cc.expose_headroom:
        used = sum(scaled.values())
        headroom['building'] = None if cc.building_limit_kw is None else cc.building_limit_kw - used
        for ph in cc.phases:
            headroom[ph['name']] = None if ph['limit_kw'] is None else ph['limit_kw'] - sum(scaled.get(cid, 0.0) for cid in ph['chargers'])
    return actions, violation_kw * dt_hours, headroom


class _FlexEnv:
    """The EVs, chargers and washing machines of ONE environment."""

    def __init__(self, spec, n_steps: int, drift: Optional[np.ndarray]):
        r = np.float64(spec.buildings[0].time_step_ratio)
        self.spec = spec
        self.n_steps = n_steps
        self.names = [ev.name for ev in spec.electric_vehicles]
        self.evs = [_EVBattery(ev.battery, r) for ev in spec.electric_vehicles]
        dt_hours = spec.seconds_per_time_step / 3600
        self.chargers = [[_Charger(c, n_steps, dt_hours) for c in b.chargers] for b in spec.buildings]
        self.wms = [[_WashingMachine(w, n_steps) for w in b.washing_machines] for b in spec.buildings]
        self.drift = drift              # [n_steps, n_ev] multipliers replayed instead of np.random.normal draws, or None
        self.drift_log = np.ones((n_steps, len(self.names)), dtype=np.float64)   # the multipliers actually used, by (step, EV)
        self.t = 0

    def all_chargers(self):
        return [c for row in self.chargers for c in row]

    def reset(self):
        self.t = 0
        for ev in self.evs:
            ev.reset()
            ev.t = 0
        for c in self.all_chargers():
            c.begin_step()
        for row in self.wms:
            for w in row:
                w.reset()
        self.associate()

    def associate(self):
        """CityLearnEnv.associate_chargers_to_electric_vehicles (citylearn.py:1353-1414)."""
        t = self.t
        for c in self.all_chargers():
            sim = c.sim
            state = sim['electric_vehicle_charger_state'][t]
            if np.isnan(state) or state not in [1, 2]:
                continue
            ev_id = sim['electric_vehicle_id'][t]
            prev_state, prev_id = np.nan, None
            if t > 0:
                prev_state = sim['electric_vehicle_charger_state'][t - 1]
                prev_id = sim['electric_vehicle_id'][t - 1]
            if isinstance(ev_id, str) and ev_id.strip() not in ['', 'nan']:
                for k, name in enumerate(self.names):
                    if name != ev_id:
                        continue
                    if state == 1:
                        if c.connected is not None:
                            raise ValueError('Charger is already in use.')
                        c.connected = k
                        new = prev_state != 1 or not isinstance(prev_id, str) or prev_id != ev_id
                        if new:
                            soc = self._arrival_soc(sim, t, prev_state, prev_id, ev_id)
                            if soc is not None:
                                self.evs[k].force_set_soc(soc)
                    elif state == 2:
                        c.incoming = k

    @staticmethod
    def _arrival_soc(sim, step, prev_state, prev_id, ev_id):
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

    def simulate_unconnected(self):
        """CityLearnEnv.simulate_unconnected_ev_soc (citylearn.py:1416-1474), called with the NEW time step."""
        t = self.t
        if t + 1 >= self.n_steps:
            return
        for k, name in enumerate(self.names):
            found = False
            for c in self.all_chargers():
                sim = c.sim
                ids, st = sim['electric_vehicle_id'], sim['electric_vehicle_charger_state']
                cur_id, nxt_id, cur_st, nxt_st = ids[t], ids[t + 1], st[t], st[t + 1]
                if isinstance(cur_id, str) and cur_id == name and cur_st == 1:
                    found = True
                    break
                connecting = isinstance(nxt_id, str) and nxt_id == name and nxt_st == 1 and cur_st != 1
                incoming = isinstance(cur_id, str) and cur_id == name and cur_st == 2
                if connecting:
                    found = True
                    arr = sim['electric_vehicle_estimated_soc_arrival']
                    soc = arr[t] if incoming else arr[t + 1]
                    if 0 <= soc <= 1:
                        self.evs[k].force_set_soc(soc)
                    break
            if not found and t > 0:
                last = self.evs[k].prev_soc
                mult = np.random.normal(1.0, 0.2) if self.drift is None else self.drift[t, k]
                self.drift_log[t, k] = mult
                variability = np.clip(mult, 0.6, 1.4)
                self.evs[k].force_set_soc(np.clip(last * variability, 0.0, 1.0))

    def next_time_step(self):
        self.t += 1
        for ev in self.evs:
            ev.begin_step(self.t)
        for c in self.all_chargers():
            c.begin_step()
        for row in self.wms:
            for w in row:
                w.begin_step(self.t)
        self.simulate_unconnected()
        self.associate()


def ev_reward(env: _FlexEnv, units, base_rewards, weights=EV_REWARD_WEIGHTS, violations=None, coefficient: float = 1.0) -> List[float]:
    """`Electric_Vehicles_Reward_Function.calculate` (reward_function.py:415-531), decentralised form.  Buildings
    without chargers get 0; the 'no_car_charging' term is computed and then dropped by the reference's `continue`."""
    t = env.t
    out = []
    for b, u in enumerate(units):
        chargers = env.chargers[b]
        if not chargers:
            out.append(0)
            continue
        current = base_rewards[b]
        mult = 1.0 / (1.0 + abs(current))
        net = u.net
        total = 0.0
        for c in chargers:
            if c.connected is None:
                continue
            ev = env.evs[c.connected]
            bs = env.spec.electric_vehicles[c.connected].battery
            soc_prev = bs.initial_soc if t == 0 else ev.prev_soc      # building.py:1355
            soc_now = ev.soc
            capacity = bs.capacity
            min_capacity = (1 - bs.depth_of_discharge) * capacity
            last = float(c.energy_kwh)
            required = c.sim['electric_vehicle_required_soc_departure'][t]
            hours = c.sim['electric_vehicle_departure_time'][t]
            k = {n: 0.0 for n in weights}
            energy = soc_prev * capacity + last
            if energy > capacity or energy < min_capacity:
                k['battery_limits'] += weights['battery_limits'] * mult
            diff = soc_now - required
            diff_kwh = diff * capacity
            max_c = c.spec.max_charging_power * hours
            max_d = c.spec.max_discharging_power * hours
            if diff_kwh > max_c:
                k['soc_impossible'] += weights['soc_impossible'] * mult
            if hours == 0:
                if -0.25 < diff <= -0.10:
                    k['soc_under'] += 2 * weights['soc_under'] * mult
                elif diff <= -0.25:
                    k['soc_under'] += (weights['soc_under'] ** 2) * mult
                elif -0.10 < diff <= 0.10:
                    k['close_soc'] += weights['close_soc'] * mult
            if abs(diff_kwh) <= max(max_c, max_d):
                k['close_soc'] += weights['close_soc'] * mult * (1.0 / (hours + 0.1))
            if last > 0 and net < 0:
                k['extra_self_production'] += weights['extra_self_production'] * mult
            elif last < 0 and net < 0:
                k['extra_self_production'] += -0.5 * weights['extra_self_production'] * mult
            if last < 0 and net > 0:
                k['self_ev_consumption'] += weights['self_ev_consumption'] * mult
            elif last > 0 and net > 0:
                k['self_ev_consumption'] += -0.5 * weights['self_ev_consumption'] * mult
            total += sum(k.values())
        out.append(total)
    if violations is not None:                     # reward_function.py:431-434
        out = [r - (v * coefficient if v > 0.0 else 0.0) for r, v in zip(out, violations)]
    return out


class FlexDistrictOracle(DistrictOracle):
    """`DistrictOracle` plus the district's EVs, chargers and washing machines.

    ``drift``: optional [n_steps, n_ev] array of the N(1, 0.2) multipliers of the unconnected-EV SoC drift
    (citylearn.py:1468-1472); without it the oracle draws them from ``np.random`` exactly like the reference, so a test
    that seeds ``np.random`` the way the fixture generator did replays the reference's stream.
    """

    def __init__(self, spec, tables, n_env: int, reward: str = 'RewardFunction', exponent: float = 1.0,
                 t0_quirk: bool = True, drift: Optional[np.ndarray] = None):
        DistrictOracle.__init__(self, spec, tables, n_env, reward, exponent, t0_quirk)
        self.flex = [_FlexEnv(spec, tables.n_steps, drift) for _ in range(n_env)]

    def reset(self):
        DistrictOracle.reset(self)
        for f in self.flex:
            f.reset()

    def step(self, actions: np.ndarray) -> Dict[str, np.ndarray]:
        B, E = len(self.spec.buildings), self.n_env
        spec = self.spec
        n_ev = len(spec.electric_vehicles)
        n_c = sum(len(b.chargers) for b in spec.buildings)
        n_w = sum(len(b.washing_machines) for b in spec.buildings)
        out = {k: np.zeros((B, E), dtype=np.float32) for k in ('net', 'reward', 'soc', 'eb', 'base_net', 'chargers_total', 'wms_total')}
        out['cc_violation_kwh'] = np.zeros((B, E), dtype=np.float64)
        out['cc_headroom'] = [[None] * E for _ in range(B)]
        out.update({k: np.zeros(E, dtype=np.float32) for k in ('d_net', 'd_cost', 'd_emission', 'd_reward')})
        out.update(ev_soc=np.zeros((n_ev, E), np.float32), ev_degcap=np.zeros((n_ev, E), np.float64),
                   ev_soc_next=np.zeros((n_ev, E), np.float32),
                   charger_consumption=np.zeros((n_c, E), np.float32), charger_energy=np.zeros((n_c, E), np.float32),
                   wm_consumption=np.zeros((n_w, E), np.float32))
        t = self.t
        with np.errstate(all='ignore'):
            for e, (env, fx) in enumerate(zip(self.units, self.flex)):
                per_b: List[Dict[str, float]] = [dict() for _ in env]
                for c, (bi, name) in enumerate(self.columns):
                    per_b[bi][name] = float(actions[c, e])
                violations = [0.0] * len(env)
                for b, (u, a) in enumerate(zip(env, per_b)):
                    u.begin_step(t)
                    if u.spec.charging_constraints is not None:     # building.py:1539-1540: before every device
                        ev_a = {ch.spec.charger_id: a[ch.spec.action_name] for ch in fx.chargers[b] if ch.spec.action_name in a}
                        ev_a, violations[b], out['cc_headroom'][b][e] = apply_charging_constraints(u.spec, ev_a, fx.chargers[b][0].dt_hours if fx.chargers[b] else 1.0)
                        a = {**a, **{ch.spec.action_name: ev_a[ch.spec.charger_id] for ch in fx.chargers[b] if ch.spec.charger_id in ev_a}}
                        out['cc_violation_kwh'][b, e] = violations[b]
                    u.apply_actions(a)
                    for ch in fx.chargers[b]:                       # building.py:1581-1592
                        if ch.spec.action_name in a:
                            ch.update(a[ch.spec.action_name], fx.evs)
                    for w in fx.wms[b]:                             # building.py:1594-1604
                        if w.spec.name in a:
                            w.start_cycle(a[w.spec.name], t)
                    total = 0
                    for ch in fx.chargers[b]:
                        total = total + ch.consumption
                    u.chargers_total = F32(total)
                    total = 0
                    for w in fx.wms[b]:
                        total = total + w.consumption * 1           # WashingMachine.time_step_ratio is 1 (energy_model.py:38)
                    u.wms_total = F32(total)
                for u in env:
                    u.update_variables()
                if self.reward == 'Electric_Vehicles_Reward_Function':
                    marl = reward_values('MARL', env)
                    if self.spec.central_agent:             # MARL returns [sum]; every building is scaled by it (reward_function.py:423-425)
                        marl = [sum(marl)] * len(env)
                    rewards = ev_reward(fx, env, marl, violations=violations)
                else:
                    rewards = reward_values(self.reward, env, self.exponent)
                out['d_net'][e] = sum(u.net for u in env)
                out['d_cost'][e] = sum(u.cost for u in env)
                out['d_emission'][e] = sum(u.emission for u in env)
                out['d_reward'][e] = sum(rewards)
                for b, u in enumerate(env):
                    out['net'][b, e] = u.net
                    out['reward'][b, e] = rewards[b]
                    out['soc'][b, e] = u.es.soc
                    out['eb'][b, e] = u.es.eb
                    out['base_net'][b, e] = u.net_without_storage()
                    out['chargers_total'][b, e] = u.chargers_total
                    out['wms_total'][b, e] = u.wms_total
                out['ev_soc'][:, e] = [ev.soc for ev in fx.evs]
                out['ev_degcap'][:, e] = [ev.degraded_capacity for ev in fx.evs]
                out['charger_consumption'][:, e] = [c.consumption for c in fx.all_chargers()]
                out['charger_energy'][:, e] = [c.energy_kwh for c in fx.all_chargers()]
                out['wm_consumption'][:, e] = [w.consumption for row in fx.wms for w in row]
                if t + 1 < self.tables.n_steps:
                    fx.next_time_step()
                    out['ev_soc_next'][:, e] = [ev.soc for ev in fx.evs]
        self.t += 1
        return out
