"""CPU oracle: scalar restatement of the CityLearn step arithmetic (TEST INFRASTRUCTURE, NOT PRODUCT CODE).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module,
and only as the checker.  Nothing under ``citylearn_amd/`` imports it.

Parity pinning: this restatement is checked against trajectories produced by running the reference itself
(`oracle/ref_harness/gen_golden.py` -> ``tests/golden/*.npz``; see ``tests/test_oracle_golden.py``) on the
2022 (battery + PV), 2020 (heat pump, chilled-water tank, heater, battery, PV) and 2023 (power outage,
partial-load cooling) schemas.  The reference has no golden vectors of its own for this path.

Every function cites the reference lines it follows (paths relative to /root/reference/citylearn/).  The
arithmetic deliberately mirrors the reference's mixed precision: state series are float32 arrays, parameters
are Python floats / float64 curve tables, so numpy's promotion rules reproduce the reference's rounding.

One `UnitOracle` = one building of one environment.  `DistrictOracle` = B buildings x E environments with the
env-level aggregation of `CityLearnEnv.step` / `update_variables` and the reward functions.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import numpy as np

ZDP = 1e-6       # data.py:19  ZERO_DIVISION_PLACEHOLDER
TOLERANCE = 1e-4  # data.py:18

F32 = np.float32


def _interp_index(x, xs) -> int:
    """`idx = max(0, np.argmax(x <= xs) - 1)` (energy_model.py:1083, 1103)."""
    return max(0, int(np.argmax(x <= xs)) - 1)


class _Tank:
    """StorageDevice / StorageTank (energy_model.py:603-870)."""

    def __init__(self, spec, r: float):
        self.capacity = spec.capacity
        self.efficiency = spec.efficiency
        self.loss_raw = spec.loss_coefficient
        self.initial_soc = spec.initial_soc
        self.max_input_power = spec.max_input_power
        self.max_output_power = spec.max_output_power
        self.r = r
        self.reset()

    def reset(self):
        # StorageDevice.reset (energy_model.py:797-803): float32 series, soc[0] = initial_soc
        self.prev_soc = F32(self.initial_soc)
        self.soc = F32(self.initial_soc)
        self.eb = F32(0.0)

    @property
    def loss_coefficient(self):          # energy_model.py:650-654
        return self.loss_raw * self.r

    @property
    def rte(self):                        # energy_model.py:674-678
        return self.efficiency ** 0.5

    def begin_step(self, t: int):
        # arrays are zero-initialised: soc[t] / energy_balance[t] read 0 until charge() writes them (t > 0)
        if t > 0:
            self.prev_soc = self.soc
            self.soc = F32(0.0)
            self.eb = F32(0.0)

    def energy_init(self):                # energy_model.py:661-666
        return max(0.0, self.prev_soc * self.capacity * (1 - self.loss_coefficient))

    def base_charge(self, energy):        # StorageDevice.charge (energy_model.py:719-739)
        energy = energy * self.r
        e_init = self.energy_init()
        e_final = min(e_init + energy * self.rte, self.capacity) if energy >= 0 \
            else max(0.0, e_init + energy / self.rte)
        self.soc = F32(e_final / max(self.capacity, ZDP))
        self.eb = F32(self.set_energy_balance(e_final, e_init))

    def set_energy_balance(self, energy, e_init):   # energy_model.py:744-768
        d = energy - e_init
        return d / self.rte if d >= 0 else d * self.rte

    def charge(self, energy):             # StorageTank.charge (energy_model.py:850-870)
        energy = energy * self.r
        if energy >= 0:
            energy = energy if self.max_input_power is None else np.nanmin([energy, self.max_input_power])
        else:
            energy = energy if self.max_output_power is None else np.nanmax([-self.max_output_power, energy])
        self.base_charge(energy)


class _Battery(_Tank):
    """Battery (energy_model.py:872-1242)."""

    def __init__(self, spec, r: float):
        self.nominal_power = spec.nominal_power
        self.clc = spec.capacity_loss_coefficient
        self.dod = spec.depth_of_discharge
        self.pec = np.array(spec.power_efficiency_curve, dtype=float)
        self.cpc = np.array(spec.capacity_power_curve, dtype=float)
        self.eff0 = spec.efficiency
        self.max_input_power = None
        self.max_output_power = None
        self.capacity = spec.capacity
        self.loss_raw = spec.loss_coefficient
        self.initial_soc = spec.initial_soc
        self.r = r
        self.reset()

    def reset(self):
        _Tank.reset(self)
        self.efficiency = self.eff0            # efficiency_history[0:1] (energy_model.py:1240)
        self.degraded_capacity = self.capacity  # capacity_history[0:1]  (energy_model.py:1241)
        self.ec = F32(0.0)                      # ElectricDevice series (energy_model.py:151-155)

    def begin_step(self, t: int):
        _Tank.begin_step(self, t)
        if t > 0:
            self.ec = F32(0.0)

    def consumption(self):                 # ElectricDevice.electricity_consumption (energy_model.py:115-118)
        return self.ec * self.r

    def get_max_input_power(self):         # energy_model.py:1070-1090
        soc = self.energy_init() / max(self.capacity, ZDP)
        xs, ys = self.cpc
        i = _interp_index(soc, xs)
        return self.nominal_power * (ys[i] + (ys[i + 1] - ys[i]) * (soc - xs[i]) / (xs[i + 1] - xs[i]))

    def get_current_efficiency(self, energy):   # energy_model.py:1092-1109
        x = np.abs(energy) / max(self.nominal_power, ZDP)
        xs, ys = self.pec
        i = _interp_index(x, xs)
        return ys[i] + (x - xs[i]) * (ys[i + 1] - ys[i]) / (xs[i + 1] - xs[i])

    def degrade(self):                     # energy_model.py:1130-1141
        d = self.clc * self.capacity * np.abs(self.eb) / (2 * max(self.degraded_capacity, ZDP))
        return d * self.r

    def charge(self, energy):              # Battery.charge (energy_model.py:1027-1057)
        energy = energy * self.r
        action_energy = energy
        if energy >= 0:
            wrt_degrade = self.degraded_capacity - self.energy_init()
            pmax = self.get_max_input_power()
            energy = min(pmax, self.nominal_power - self.consumption(), wrt_degrade, energy)
            self.efficiency = self.get_current_efficiency(min(action_energy, pmax))
        else:
            soc_limit = 1.0 - self.dod
            limit = max((self.prev_soc - soc_limit) * self.capacity * self.rte, 0.0) * -1
            pmax = self.get_max_input_power()
            energy = max(-pmax, limit, energy)
            self.efficiency = self.get_current_efficiency(min(abs(action_energy), pmax))
        self.base_charge(energy)
        self.degraded_capacity = max(self.degraded_capacity - self.degrade(), 0.0)
        self.ec = F32(self.ec + self.eb)       # update_electricity_consumption(enforce_polarity=False)


class UnitOracle:
    """One building of one environment: `Building.apply_actions` .. `update_variables` (building.py)."""

    def __init__(self, bspec, tables, b_index: int, t0_quirk: bool = True):
        self.spec = bspec
        self.b = b_index
        self.r = np.float64(bspec.time_step_ratio)   # np.float64 in the reference: py number / np.int32 (data.py:427-450)
        self.dt = bspec.seconds_per_time_step / 3600          # building.py:113
        self.t0_quirk = t0_quirk
        w = slice(tables.start, tables.end + 1)
        s = bspec.series
        self.nsl = s['non_shiftable_load'][w]
        self.cool_dem_ideal = s['cooling_demand'][w]
        self.heat_dem_ideal = s['heating_demand'][w]
        self.dhw_dem = s['dhw_demand'][w]
        self.t_out = s['outdoor_dry_bulb_temperature'][w]
        self.price = s['electricity_pricing'][w]
        self.carbon = s['carbon_intensity'][w]
        self.hvac_mode = s['hvac_mode'][w]
        # building.py:2554  -pv.get_generation(W/kW)  (energy_model.py:469-488)
        self.solar = (bspec.pv_nominal_power * np.array(s['solar_generation'][w]) / 1000.0) * -1
        self.outage_signal = tables.outage[:, b_index]
        self.simulate_outage = bspec.outage.simulate
        self.cs = _Tank(bspec.cooling_storage, self.r)
        self.hs = _Tank(bspec.heating_storage, self.r)
        self.ds = _Tank(bspec.dhw_storage, self.r)
        self.es = _Battery(bspec.electrical_storage, self.r)
        # chargers / washing machines of this building at the current step (oracle/flex_oracle.py sets them; 0 otherwise)
        self.chargers_total = F32(0.0)
        self.wms_total = F32(0.0)
        self.reset()

    # -- helpers ------------------------------------------------------------------------------------------
    def cop(self, dev, heating: bool):
        """HeatPump.get_cop (energy_model.py:216-250) or the heater's efficiency."""
        if not dev.is_heat_pump:
            return dev.efficiency
        return dev.cop(self.t_out[self.t], heating)

    def outage(self) -> bool:              # building.py:671-674
        return self.simulate_outage and bool(self.outage_signal[self.t])

    def flex(self):                        # Building.downward_electrical_flexibility (building.py:640-668)
        if not self.outage():
            return np.inf
        r = self.r
        cap = abs(self.solar[self.t]) - (self.c_cool * r + self.c_heat * r + self.c_dhw * r + self.c_ns * r
                                         + self.es.consumption())
        return max(0.0, cap)

    def max_out(self, dev, c, heating: bool):
        """get_max_output_power(max_electric_power=flex) (energy_model.py:252-281, 378-401)."""
        avail = dev.nominal_power - c * self.r      # available_nominal_power (energy_model.py:120-124)
        return np.min([self.flex(), avail], axis=0) * self.cop(dev, heating)

    # -- episode ------------------------------------------------------------------------------------------
    def reset(self):
        """Building.reset (building.py:2526-2564) followed by the reset-time update_variables (citylearn.py:1884)."""
        self.t = 0
        for d in (self.cs, self.hs, self.ds, self.es):
            d.reset()
        self.c_cool = F32(0.0); self.c_heat = F32(0.0); self.c_dhw = F32(0.0); self.c_ns = F32(0.0)
        self.cool_dem = F32(self.cool_dem_ideal[0]); self.heat_dem = F32(self.heat_dem_ideal[0])
        # energy_from_*_device / energy_to_non_shiftable_load start as copies of the demand series (2555-2558)
        self.e_cool_dev = F32(self.cool_dem_ideal[0]); self.e_heat_dev = F32(self.heat_dem_ideal[0])
        self.e_dhw_dev = F32(self.dhw_dem[0]); self.e_ns = F32(self.nsl[0])
        self.net = F32(0.0); self.cost = F32(0.0); self.emission = F32(0.0)
        self.update_variables()

    def begin_step(self, t: int):
        """Positions the zero-initialised per-step slots (arrays of zeros in the reference)."""
        self.t = t
        if t > 0:
            self.c_cool = F32(0.0); self.c_heat = F32(0.0); self.c_dhw = F32(0.0); self.c_ns = F32(0.0)
            self.cool_dem = F32(self.cool_dem_ideal[t]); self.heat_dem = F32(self.heat_dem_ideal[t])
            self.e_cool_dev = F32(self.cool_dem_ideal[t]); self.e_heat_dev = F32(self.heat_dem_ideal[t])
            self.e_dhw_dev = F32(self.dhw_dem[t]); self.e_ns = F32(self.nsl[t])
        for d in (self.cs, self.hs, self.ds, self.es):
            d.begin_step(t)

    # -- apply_actions (building.py:1500-1634) ---------------------------------------------------------------
    def apply_actions(self, actions: Dict[str, float]):
        sp = self.spec
        active = sp.active_actions
        a_cd = actions.get('cooling_device', np.nan)
        a_hd = actions.get('heating_device', np.nan)
        if 'cooling_or_heating_device' in active:
            a = actions['cooling_or_heating_device']
            a_cd, a_hd = abs(min(a, 0.0)), abs(max(a, 0.0))
        else:
            a_cd = np.nan if 'cooling_device' not in active else a_cd
            a_hd = np.nan if 'heating_device' not in active else a_hd
        a_cs = 0.0 if 'cooling_storage' not in active else actions['cooling_storage']
        a_hs = 0.0 if 'heating_storage' not in active else actions['heating_storage']
        a_ds = 0.0 if 'dhw_storage' not in active else actions['dhw_storage']
        a_es = 0.0 if 'electrical_storage' not in active else actions['electrical_storage']
        order = ['cooling_demand', 'heating_demand', 'cooling_device', 'cooling_storage', 'heating_device',
                 'heating_storage', 'dhw_device', 'dhw_storage', 'non_shiftable_load', 'electrical_storage']
        if a_es < 0.0:
            order.remove('electrical_storage')
            order = ['electrical_storage'] + order
        for key, a in (('cooling', a_cs), ('heating', a_hs), ('dhw', a_ds)):
            if a < 0.0:
                i, j = order.index(f'{key}_storage'), order.index(f'{key}_device')
                order[i], order[j] = f'{key}_device', f'{key}_storage'
        fn = {
            'cooling_demand': lambda: self.update_cooling_demand(a_cd),
            'heating_demand': lambda: self.update_heating_demand(a_hd),
            'cooling_device': lambda: self.update_energy_from_device('cooling'),
            'heating_device': lambda: self.update_energy_from_device('heating'),
            'dhw_device': lambda: self.update_energy_from_device('dhw'),
            'cooling_storage': lambda: self.update_storage('cooling', a_cs),
            'heating_storage': lambda: self.update_storage('heating', a_hs),
            'dhw_storage': lambda: self.update_storage('dhw', a_ds),
            'non_shiftable_load': self.update_non_shiftable_load,
            'electrical_storage': lambda: self.update_electrical_storage(a_es),
        }
        for k in order:
            fn[k]()

    def dynamics_active(self) -> bool:
        """LSTMDynamicsBuilding.simulate_dynamics (building.py:2996-2999): true once lookback+1 inputs exist,
        i.e. from step index lookback+1 (one `_update_dynamics_input` per completed step, building.py:3057)."""
        return self.spec.is_dynamics and self.t >= self.spec.dynamics.lookback + 1

    def update_cooling_demand(self, action):        # building.py:3080-3121 (base class: NotImplementedError)
        sp = self.spec
        if not sp.is_dynamics:
            return
        if ('cooling_device' in sp.active_actions or 'cooling_or_heating_device' in sp.active_actions) and self.dynamics_active():
            if self.hvac_mode[self.t] in (1, 3):
                power = action * sp.cooling_device.nominal_power * self.dt
                avail = sp.cooling_device.nominal_power - self.c_cool * self.r
                demand = np.min([power, avail], axis=0) * self.cop(sp.cooling_device, False)
            else:
                demand = 0.0
            self.cool_dem = F32(demand)

    def update_heating_demand(self, action):        # building.py:3123-3158
        sp = self.spec
        if not sp.is_dynamics:
            return
        if ('heating_device' in sp.active_actions or 'cooling_or_heating_device' in sp.active_actions) and self.dynamics_active():
            if self.hvac_mode[self.t] in (2, 3):
                power = action * sp.heating_device.nominal_power
                avail = sp.heating_device.nominal_power - self.c_heat * self.r
                demand = np.min([power, avail], axis=0) * self.cop(sp.heating_device, True)
            else:
                demand = 0.0
            self.heat_dem = F32(demand)

    def _end_use(self, key):
        sp = self.spec
        if key == 'cooling':
            return sp.cooling_device, self.cs, False, self.cool_dem
        if key == 'heating':
            return sp.heating_device, self.hs, True, self.heat_dem
        return sp.dhw_device, self.ds, True, F32(self.dhw_dem[self.t])

    def _add(self, key, value):
        """ElectricDevice.update_electricity_consumption: in-place float32 accumulate (energy_model.py:148)."""
        if key == 'cooling':
            self.c_cool = F32(self.c_cool + value)
        elif key == 'heating':
            self.c_heat = F32(self.c_heat + value)
        else:
            self.c_dhw = F32(self.c_dhw + value)

    def _c(self, key):
        return {'cooling': self.c_cool, 'heating': self.c_heat, 'dhw': self.c_dhw}[key]

    def update_energy_from_device(self, key):       # building.py:1641-1661, 1694-1709, 1739-1754
        dev, tank, heating, demand = self._end_use(key)
        storage_output = np.clip(tank.eb, None, 0) * -1          # energy_from_*_storage (building.py:525-541)
        max_device_output = self.max_out(dev, self._c(key), heating)
        device_output = min(demand - storage_output, max_device_output)
        out32 = F32(device_output)
        if key == 'cooling':
            self.e_cool_dev = out32
        elif key == 'heating':
            self.e_heat_dev = out32
        else:
            self.e_dhw_dev = out32
        consumption = device_output / self.cop(dev, heating)
        self._add(key, max(0.0, consumption))

    def update_storage(self, key, action):          # building.py:1663-1687, 1711-1737, 1756-1782
        dev, tank, heating, demand = self._end_use(key)
        if key == 'cooling':
            energy = action * self.cs.capacity
        elif key == 'heating':
            energy = action * self.cs.capacity * self.dt          # sic (building.py:1720)
        else:
            energy = action * self.hs.capacity * self.dt          # sic (building.py:1765)
        if energy > 0.0:
            energy = min(self.max_out(dev, self._c(key), heating), energy)
        else:
            energy = max(-demand, energy)
        tank.charge(energy / self.r if self.r not in (None, 0) else energy)   # _convert_energy_for_storage (1814-1823)
        charged = max(tank.eb, 0.0)
        self._add(key, charged / self.cop(dev, heating))

    def update_non_shiftable_load(self):            # building.py:1784-1789
        demand = min(self.nsl[self.t], self.flex())
        self.e_ns = F32(demand)
        self.c_ns = F32(self.c_ns + demand)

    def update_electrical_storage(self, action):    # building.py:1791-1812
        power = action * self.es.nominal_power
        energy = power * (self.spec.seconds_per_time_step / 3600)
        energy = min(energy, self.flex())
        self.es.charge(energy / self.r if self.r not in (None, 0) else energy)

    # -- update_variables (building.py:2615-2703) -------------------------------------------------------------
    def update_variables(self):
        sp = self.spec
        r = self.r
        if self.t == 0 and self.t0_quirk:
            # executed at reset AND again in the first step because time_step is still 0
            self.c_cool = F32(self.c_cool + (self.e_cool_dev + self.cs.eb) / self.cop(sp.cooling_device, False))
            hd = (self.e_heat_dev + self.hs.eb)
            if sp.heating_device.is_heat_pump:
                self.c_heat = F32(self.c_heat + hd / self.cop(sp.heating_device, True))
            else:
                self.c_heat = F32(self.c_heat + np.array(hd) / sp.dhw_device.efficiency)   # sic (building.py:2632)
            self.c_dhw = F32(self.c_dhw + (self.e_dhw_dev + self.ds.eb) / self.cop(sp.dhw_device, True))
            self.c_ns = F32(self.c_ns + self.e_ns)
            self.es.ec = F32(self.es.ec + self.es.eb)
        net = 0.0
        if not self.outage():
            net = self.c_cool * r + self.c_heat * r + self.c_dhw * r + self.c_ns * r + self.es.consumption() \
                + self.solar[self.t] + self.chargers_total + self.wms_total     # building.py:2685-2693
        self.net = F32(net)
        self.cost = F32(net * self.price[self.t])
        self.emission = F32(max(0.0, net * self.carbon[self.t]))

    # -- quantities read by rewards / KPIs --------------------------------------------------------------------
    def delivered_cooling(self):            # reward observation 'cooling_demand' (building.py:1435)
        return self.e_cool_dev + abs(min(self.cs.eb, 0.0))

    def storage_electricity(self, key):     # *_storage_electricity_consumption (building.py:413-457)
        dev, tank, heating, _ = self._end_use(key)
        return tank.eb / self.cop(dev, heating)

    def net_without_storage(self):          # building.py:345-366 at the current step
        return self.net - np.sum([self.storage_electricity('cooling'), self.storage_electricity('heating'),
                                  self.storage_electricity('dhw'), self.es.consumption(), self.chargers_total], axis=0)

    def net_without_storage_and_partial_load(self):   # building.py:2877-2905 at the current step
        sp = self.spec
        dc = (self.cool_dem_ideal[self.t] - self.cool_dem) / self.cop(sp.cooling_device, False)
        dh = self.heat_dem_ideal[self.t] - self.heat_dem
        # sic (building.py:2893-2898): the heating difference of EVERY step is converted with the heat pump's COP at the outdoor
        # temperature of the step the series is read at -- the episode's last one when evaluate() runs after the episode
        dh = dh / sp.heating_device.cop(self.t_out[len(self.t_out) - 1], True) if sp.heating_device.is_heat_pump else np.array(dh) / sp.dhw_device.efficiency
        return self.net_without_storage() + np.sum([dc, dh], axis=0)


# --------------------------------------------------------------------------------------------------------------
# rewards (reward_function.py)
# --------------------------------------------------------------------------------------------------------------

def reward_values(kind: str, units: Sequence[UnitOracle], exponent: float = 1.0) -> List[float]:
    net = [u.net for u in units]
    if kind == 'RewardFunction':                      # reward_function.py:65-88
        return [-(max(o, 0) ** exponent) for o in net]
    if kind == 'IndependentSACReward':                # reward_function.py:159-168 (`v*-1**3` parses as -v)
        return [min(v * -1 ** 3, 0) for v in net]
    if kind == 'MARL':                                # reward_function.py:132-143
        district = sum(net)
        b = np.array(net, dtype=float) * -1
        return (np.sign(b) * 0.01 * b ** 2 * np.nanmax([0, district])).tolist()
    if kind == 'SolarPenaltyReward':                  # reward_function.py:189-214
        out = []
        for u in units:
            e = u.net
            rew = 0.0
            for cap, soc in ((u.cs.capacity, u.cs.soc), (u.hs.capacity, u.hs.soc), (u.ds.capacity, u.ds.soc),
                             (u.es.capacity, u.es.soc)):
                rew += -(1.0 + np.sign(e) * soc) * abs(e) if cap > ZDP else 0.0
            out.append(rew)
        return out
    raise NotImplementedError(kind)


class DistrictOracle:
    """B buildings x E environments: `CityLearnEnv.reset/step` aggregation (citylearn.py:978-1056, 1829-1918)."""

    def __init__(self, spec, tables, n_env: int, reward: str = 'RewardFunction', exponent: float = 1.0,
                 t0_quirk: bool = True):
        self.spec, self.tables, self.n_env = spec, tables, n_env
        self.reward, self.exponent = reward, exponent
        self.units = [[UnitOracle(b, tables, i, t0_quirk) for i, b in enumerate(spec.buildings)] for _ in range(n_env)]
        self.columns = spec.action_columns
        self.t = 0

    def reset(self):
        self.t = 0
        for env in self.units:
            for u in env:
                u.reset()

    def step(self, actions: np.ndarray) -> Dict[str, np.ndarray]:
        """actions: [n_act_cols, n_env].  Returns per-step arrays (float32) for parity checks."""
        B, E = len(self.spec.buildings), self.n_env
        out = {k: np.zeros((B, E), dtype=np.float32) for k in
               ('net', 'reward', 'soc', 'eff', 'degcap', 'eb', 'cs_soc', 'hs_soc', 'ds_soc', 'c_cool', 'c_heat',
                'c_dhw', 'c_ns', 'cool_dem', 'base_net')}
        out.update({k: np.zeros(E, dtype=np.float32) for k in ('d_net', 'd_cost', 'd_emission', 'd_reward')})
        with np.errstate(all='ignore'):
            for e, env in enumerate(self.units):
                per_b: List[Dict[str, float]] = [dict() for _ in env]
                for c, (bi, name) in enumerate(self.columns):
                    per_b[bi][name] = float(actions[c, e])   # agents hand Python floats to env.step
                for u, a in zip(env, per_b):
                    u.begin_step(self.t)
                    u.apply_actions(a)
                for u in env:
                    u.update_variables()
                rewards = reward_values(self.reward, env, self.exponent)
                # district sums: python sum() of float32 scalars, building order (citylearn.py:1909-1918)
                out['d_net'][e] = sum(u.net for u in env)
                out['d_cost'][e] = sum(u.cost for u in env)
                out['d_emission'][e] = sum(u.emission for u in env)
                out['d_reward'][e] = sum(rewards)
                for b, u in enumerate(env):
                    out['net'][b, e] = u.net
                    out['reward'][b, e] = rewards[b]
                    out['soc'][b, e] = u.es.soc
                    out['eff'][b, e] = u.es.efficiency
                    out['degcap'][b, e] = u.es.degraded_capacity
                    out['eb'][b, e] = u.es.eb
                    out['cs_soc'][b, e] = u.cs.soc
                    out['hs_soc'][b, e] = u.hs.soc
                    out['ds_soc'][b, e] = u.ds.soc
                    # what `Device.electricity_consumption` reports: the accumulator times time_step_ratio (energy_model.py:118)
                    out['c_cool'][b, e] = u.c_cool * u.r
                    out['c_heat'][b, e] = u.c_heat * u.r
                    out['c_dhw'][b, e] = u.c_dhw * u.r
                    out['c_ns'][b, e] = u.c_ns * u.r
                    out['cool_dem'][b, e] = u.delivered_cooling()
                    out['base_net'][b, e] = u.net_without_storage_and_partial_load() if u.spec.is_dynamics \
                        else u.net_without_storage()
        self.t += 1
        return out
