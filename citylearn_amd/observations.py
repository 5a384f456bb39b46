"""Observation layout of a district: names, limits, env-independent tables and the device column map.

Host-side mirror (behaviour, not code) of

* ``Building.observations`` / ``_get_observations_data``        /root/reference/citylearn/building.py:1115-1219, 1336-1481
* ``Building.estimate_observation_space(_limits)``              /root/reference/citylearn/building.py:1836-2158
* ``CityLearnEnv.observation_names / observation_space / observations`` (central-agent de-duplication of shared
  observations)                                                 /root/reference/citylearn/citylearn.py:330-485
* ``NormalizedObservationWrapper`` (periodic sin / cos + min-max) /root/reference/citylearn/wrappers.py:39-167,
  ``PeriodicNormalization`` / ``Normalize``                     /root/reference/citylearn/preprocessing.py:37-152

Of the ~28 active observations per building only a handful depend on the environment's own trajectory (storage
SoCs, net / device electricity consumption, delivered demand, predicted indoor temperature); everything else is a
pure function of the data files and the time step.  This module therefore produces, per episode,

* ``table [T, N]``: the env-independent value of every observation column (already normalised if asked) and, for
  env-dependent columns, the additive offset of the affine map below;
* ``col_src [N]`` / ``col_scale [N]``: for env-dependent columns the device plane the value comes from and the
  multiplicative part of ``obs = plane * scale + table[row]``;

which `cl_observe_f32` (csrc/cl_observe.h) turns into the ``[n_env, N]`` observation tensor in one write-bound pass.

Two observation semantics: ``'reference'`` -- what the reference returns: it reads the series of
time step t+1 *before* that step is simulated, so env-dependent observations are the untouched (zero) slots, except
at reset; ``'current'`` -- exogenous values of t+1 with the env-dependent values just computed at t.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import abi
from .schema import BuildingSpec, DistrictSpec, EpisodeTables

PERIODIC = {'hour': 24, 'day_type': 7, 'month': 12, 'minutes': 60}     # max of the ranges (building.py:1493-1498)
_PERIODIC_RANGE = {'hour': range(1, 25), 'day_type': range(1, 8), 'month': range(1, 13), 'minutes': range(1, 61)}

MAXIMUM_TEMPERATURE_DELTA = 20.0          # building.py:1022
OBSERVATION_SPACE_LIMIT_DELTA = 0.0       # building.py:1012
DEMAND_OBSERVATION_LIMIT_FACTOR = 2.0     # building.py:1032 (the setter's default, not the docstring's)

# kinds of device source (col_src = kind << 28 | plane << 20 | building)
SRC_STATE, SRC_OUT, SRC_TEMP, SRC_EXTRA = 0, 1, 2, 3      # SRC_EXTRA: cl_flex.flex_out planes, row = flexible-load row of the building

# env-dependent observation -> (kind, plane) for every building
_DEVICE_SOURCE = {
    'electrical_storage_soc': (SRC_STATE, abi.CLS_B_SOC), 'cooling_storage_soc': (SRC_STATE, abi.CLS_CS_SOC),
    'heating_storage_soc': (SRC_STATE, abi.CLS_HS_SOC), 'dhw_storage_soc': (SRC_STATE, abi.CLS_DS_SOC),
    'net_electricity_consumption': (SRC_OUT, abi.CLO_NET),
    'cooling_electricity_consumption': (SRC_OUT, abi.CLO_C_COOL), 'heating_electricity_consumption': (SRC_OUT, abi.CLO_C_HEAT),
    'dhw_electricity_consumption': (SRC_OUT, abi.CLO_C_DHW), 'electrical_storage_electricity_consumption': (SRC_OUT, abi.CLO_B_EB),
    'cooling_demand': (SRC_OUT, abi.CLO_COOL_DEM), 'heating_demand': (SRC_OUT, abi.CLO_HEAT_DEM), 'dhw_demand': (SRC_OUT, abi.CLO_DHW_DEM),
    'cooling_storage_electricity_consumption': (SRC_OUT, abi.CLO_SE_COOL), 'heating_storage_electricity_consumption': (SRC_OUT, abi.CLO_SE_HEAT),
    'dhw_storage_electricity_consumption': (SRC_OUT, abi.CLO_SE_DHW),
}
# planes only written with CLD_WRITE_DETAIL
_DETAIL_PLANES = {abi.CLO_C_COOL, abi.CLO_C_HEAT, abi.CLO_C_DHW, abi.CLO_B_EB, abi.CLO_COOL_DEM, abi.CLO_HEAT_DEM, abi.CLO_DHW_DEM,
                  abi.CLO_SE_COOL, abi.CLO_SE_HEAT, abi.CLO_SE_DHW}
_NO_DEVICE_PLANE = ('washing_machine_electricity_consumption',)      # the flexible-load planes carry chargers + washing machines together
ENV_DEPENDENT = set(_DEVICE_SOURCE) | set(_NO_DEVICE_PLANE)


def available_observations(b: BuildingSpec) -> set:
    """Keys of `Building._get_observations_data` (building.py:1402-1457) for a building without EVs / washing machines."""
    keys = {k for k, v in b.series.items() if isinstance(v, np.ndarray)}
    keys |= {'solar_generation', 'cooling_storage_soc', 'heating_storage_soc', 'dhw_storage_soc', 'electrical_storage_soc',
             'cooling_demand', 'heating_demand', 'dhw_demand', 'net_electricity_consumption', 'cooling_electricity_consumption',
             'heating_electricity_consumption', 'dhw_electricity_consumption', 'cooling_storage_electricity_consumption',
             'heating_storage_electricity_consumption', 'dhw_storage_electricity_consumption',
             'electrical_storage_electricity_consumption', 'washing_machine_electricity_consumption',
             'cooling_device_efficiency', 'heating_device_efficiency', 'dhw_device_efficiency',
             'indoor_dry_bulb_temperature_cooling_set_point', 'indoor_dry_bulb_temperature_heating_set_point',
             'indoor_dry_bulb_temperature_cooling_delta', 'indoor_dry_bulb_temperature_heating_delta', 'comfort_band',
             'occupant_count', 'power_outage'}
    return keys | set(flexible_load_observations(b)) | set(charging_constraint_observations(b))


def charging_constraint_observations(b: BuildingSpec) -> Dict[str, float]:
    """Observation -> value right after reset() of the charging-constraint observations (building.py:1462-1481):
    headroom = the limit, violation = 0, phase one-hot = constant."""
    cc = b.charging_constraints
    if cc is None:
        return {}
    out = dict(cc.one_hot_keys([c.charger_id for c in b.chargers]))
    out.update(dict(cc.headroom_keys()))
    if cc.expose_violation:
        out['charging_constraint_violation_kwh'] = 0.0
    return out


def flexible_load_observations(b: BuildingSpec) -> List[str]:
    """Per-charger / per-washing-machine observation names (`update_ev_charger_observations`,
    `update_washing_machine_observations`, building.py:1221-1334); values come from `flex.FlexTables.observations`."""
    out: List[str] = []
    for c in b.chargers:
        i = c.charger_id
        out += [f'electric_vehicle_charger_{i}_connected_state', f'connected_electric_vehicle_at_charger_{i}_departure_time',
                f'connected_electric_vehicle_at_charger_{i}_required_soc_departure', f'connected_electric_vehicle_at_charger_{i}_soc',
                f'connected_electric_vehicle_at_charger_{i}_battery_capacity', f'electric_vehicle_charger_{i}_incoming_state',
                f'incoming_electric_vehicle_at_charger_{i}_estimated_arrival_time',
                f'incoming_electric_vehicle_at_charger_{i}_estimated_soc_arrival']
    for w in b.washing_machines:
        out += [f'{w.name}_start_time_step', f'{w.name}_end_time_step']
    return out


def building_observation_names(b: BuildingSpec) -> List[str]:
    """Active observations in the order `Building.observations` returns them (building.py:1146-1158): the keys of
    `_get_observations_data` in metadata order, then the charger and washing-machine keys the two update_* helpers append."""
    available = available_observations(b)
    appended = flexible_load_observations(b)
    active = [k for k in b.active_observations if k in available]
    return [k for k in active if k not in appended] + [k for k in appended if k in active]


def building_space_names(b: BuildingSpec) -> List[str]:
    """Order of `Building.observation_space` (`estimate_observation_space`, building.py:1836-1865): plain metadata order --
    it differs from the order of the returned values when charging-constraint observations exist."""
    available = available_observations(b)
    return [k for k in b.active_observations if k in available]


def periodic_names(names: Sequence[str]) -> List[str]:
    """Names after `periodic_normalization` (building.py:1191-1201): ``k`` -> ``k_cos, k_sin``."""
    out: List[str] = []
    for k in names:
        out += [f'{k}_cos', f'{k}_sin'] if k in PERIODIC else [k]
    return out


def _cop_series(dev, t_out: np.ndarray, heating: bool):
    return dev.cop(t_out, heating=heating)


def space_limits(spec: DistrictSpec, b: BuildingSpec, names: Sequence[str], periodic: bool) -> Tuple[Dict[str, float], Dict[str, float]]:
    """`Building.estimate_observation_space_limits` (building.py:1867-2106) for the observation `names`, over the whole
    simulation period, with the reference's float32-series / Python-float promotion."""
    w = slice(spec.simulation_start_time_step, spec.simulation_end_time_step + 1)
    s = b.series
    t_out = s['outdoor_dry_bulb_temperature'][w]
    gen = b.pv_nominal_power * np.array(s['solar_generation'][w]) / 1000.0            # energy_model.py:488
    nsl = s['non_shiftable_load'][w]
    es, cd, hd, dd = b.electrical_storage, b.cooling_device, b.heating_device, b.dhw_device
    low: Dict[str, float] = {}
    high: Dict[str, float] = {}

    def input_power(dev, demand, heating):
        if dev.is_heat_pump:
            with np.errstate(divide='ignore', invalid='ignore'):
                return demand / dev.cop(t_out, heating=heating)
        return np.array(demand) / dev.efficiency

    flex_names = set(flexible_load_observations(b))
    cc_names = charging_constraint_observations(b)
    for key in names:
        if key in cc_names:
            # building.py:1908-1916, 2139-2156
            if key.startswith('charging_phase_one_hot_'):
                low[key], high[key] = 0.0, 1.0
            elif key == 'charging_constraint_violation_kwh':
                low[key], high[key] = 0.0, sum(c.max_charging_power or 0.0 for c in b.chargers) * (spec.seconds_per_time_step / 3600)
            else:
                low[key] = high[key] = np.float32(cc_names[key])        # min / max of a constant float32 series
        elif key in flex_names:
            # building.py:1968-2010: matched by substrings of the expanded names
            if 'connected_state' in key or '_incoming_state' in key:
                low[key], high[key] = 0, 1
            elif '_departure_time' in key or '_estimated_arrival_time' in key:
                low[key], high[key] = -1, 24
            elif '_soc' in key and '_electric_vehicle' in key:
                low[key], high[key] = -0.1, 1.0
            elif key.endswith('_battery_capacity'):
                low[key], high[key] = -1, 100
            else:                                   # washing machine start / end step
                low[key], high[key] = -1, 24
        elif key == 'net_electricity_consumption':
            lo = nsl - (+es.nominal_power + gen)
            hi = nsl + cd.nominal_power + hd.nominal_power + dd.nominal_power + es.nominal_power - gen
            low[key], high[key] = min(lo.min(), 0.0), hi.max()
        elif key.endswith('_storage_soc'):
            low[key], high[key] = 0.0, 1.0
        elif key == 'cooling_device_efficiency':
            cop = cd.cop(t_out, heating=False)
            low[key], high[key] = min(cop), max(cop)
        elif key in ('heating_device_efficiency', 'dhw_device_efficiency'):
            dev = hd if key.startswith('heating') else dd
            if dev.is_heat_pump:
                cop = dev.cop(t_out, heating=True)
                low[key], high[key] = min(cop), max(cop)
            else:
                low[key] = high[key] = dev.efficiency
        elif key == 'indoor_dry_bulb_temperature':
            x = s['indoor_dry_bulb_temperature'][w]
            low[key], high[key] = x.min() - MAXIMUM_TEMPERATURE_DELTA, x.max() + MAXIMUM_TEMPERATURE_DELTA
        elif key in ('indoor_dry_bulb_temperature_cooling_delta', 'indoor_dry_bulb_temperature_heating_delta'):
            low[key], high[key] = -MAXIMUM_TEMPERATURE_DELTA, MAXIMUM_TEMPERATURE_DELTA
        elif key == 'comfort_band':
            low[key], high[key] = 0, max(s[key][w])
        elif key in ('cooling_demand', 'heating_demand', 'dhw_demand'):
            low[key], high[key] = 0.0, s[key][w].max() * DEMAND_OBSERVATION_LIMIT_FACTOR
        elif key in ('cooling_electricity_consumption', 'heating_electricity_consumption', 'dhw_electricity_consumption'):
            dev = {'cooling': cd, 'heating': hd, 'dhw': dd}[key.split('_')[0]]
            low[key], high[key] = 0.0, dev.nominal_power
        elif key in ('cooling_storage_electricity_consumption', 'heating_storage_electricity_consumption',
                     'dhw_storage_electricity_consumption'):
            end_use = key.split('_')[0]
            dev = {'cooling': cd, 'heating': hd, 'dhw': dd}[end_use]
            low[key] = -max(input_power(dev, s[f'{end_use}_demand'][w], end_use != 'cooling'))
            high[key] = dev.nominal_power
        elif key == 'electrical_storage_electricity_consumption':
            low[key], high[key] = -es.nominal_power, es.nominal_power
        elif key == 'power_outage':
            low[key], high[key] = 0.0, 1.0
        elif periodic and key in PERIODIC:
            x = 2 * np.pi * np.array(list(_PERIODIC_RANGE[key])) / PERIODIC[key]
            low[f'{key}_cos'], high[f'{key}_cos'] = min(np.cos(x)), max(np.cos(x))
            low[f'{key}_sin'], high[f'{key}_sin'] = min(np.sin(x)), max(np.sin(x))
        elif key == 'solar_generation':
            low[key], high[key] = min(gen), max(gen)
        else:
            low[key], high[key] = min(s[key][w]), max(s[key][w])
    low = {k: v - OBSERVATION_SPACE_LIMIT_DELTA for k, v in low.items()}
    high = {k: v + OBSERVATION_SPACE_LIMIT_DELTA for k, v in high.items()}
    return low, high


class ObservationLayout:
    """Static column structure of a district's observations (`mode` / `normalize` fixed at construction)."""

    def __init__(self, spec: DistrictSpec, mode: str = 'reference', normalize: bool = False, reference_quirks: bool = True,
                 central_agent: Optional[bool] = None):
        if mode not in ('reference', 'current'):
            raise ValueError("observation mode must be 'reference' or 'current'")
        self.spec, self.mode, self.normalize, self.reference_quirks = spec, mode, normalize, reference_quirks
        self.central_agent = spec.central_agent if central_agent is None else central_agent
        self.raw_names = [building_observation_names(b) for b in spec.buildings]
        self.building_names = [periodic_names(n) if normalize else list(n) for n in self.raw_names]
        shared = list(spec.shared_observations)
        self.shared = periodic_names(shared) if normalize else shared
        # flat column list: (building, name); central agent keeps shared observations of the first building only
        # (citylearn.py:462-480 / wrappers.py:146-160)
        self.columns: List[Tuple[int, str]] = []
        self.agent_slices: List[slice] = []
        if self.central_agent:
            seen: List[str] = []
            for i, names in enumerate(self.building_names):
                for k in names:
                    if i == 0 or k not in self.shared or k not in seen:
                        self.columns.append((i, k))
                    if k in self.shared and k not in seen:
                        seen.append(k)
            self.agent_slices = [slice(0, len(self.columns))]
        else:
            for i, names in enumerate(self.building_names):
                self.agent_slices.append(slice(len(self.columns), len(self.columns) + len(names)))
                self.columns += [(i, k) for k in names]
        self._limits = [space_limits(spec, b, n, periodic=normalize) for b, n in zip(spec.buildings, self.raw_names)]
        # `observation_space` follows the metadata order of every building (citylearn.py:399-420), which differs from the
        # order of the returned values for buildings with charging-constraint observations
        self.space_columns: List[Tuple[int, str]] = []
        space_names = [periodic_names(building_space_names(b)) if normalize else building_space_names(b) for b in spec.buildings]
        if self.central_agent:
            seen = []
            for i, names in enumerate(space_names):
                for k in names:
                    if i == 0 or k not in self.shared or k not in seen:
                        self.space_columns.append((i, k))
                    if k in self.shared and k not in seen:
                        seen.append(k)
        else:
            self.space_columns = [(i, k) for i, names in enumerate(space_names) for k in names]

    @property
    def n_cols(self) -> int:
        return len(self.columns)

    @property
    def names(self) -> List[List[str]]:
        return [[k for _, k in self.columns[s]] for s in self.agent_slices]

    def limits(self) -> Tuple[np.ndarray, np.ndarray]:
        """Per-column (low, high) of the un-normalised observation (float64)."""
        lo = np.array([self._limits[i][0][k] for i, k in self.columns], dtype=np.float64)
        hi = np.array([self._limits[i][1][k] for i, k in self.columns], dtype=np.float64)
        return lo, hi

    def space(self) -> List[Tuple[np.ndarray, np.ndarray]]:
        """(low, high) float32 per agent: `CityLearnEnv.observation_space` (citylearn.py:385-425), or all [0, 1] for
        the normalised view (building.py:1856-1859)."""
        lo = np.array([self._limits[i][0][k] for i, k in self.space_columns], dtype=np.float64)
        hi = np.array([self._limits[i][1][k] for i, k in self.space_columns], dtype=np.float64)
        if self.normalize:
            lo, hi = np.zeros_like(lo), np.ones_like(hi)
        return [(lo[s].astype('float32'), hi[s].astype('float32')) for s in self.agent_slices]

    # ---- per-episode tables ------------------------------------------------------------------------------------
    def _raw_column(self, i: int, k: str, tab: EpisodeTables) -> Tuple[np.ndarray, Optional[Tuple[int, int]], np.ndarray]:
        """For raw observation `k` of building `i`: (values [T] float64 used when the column is exogenous or stale,
        device source or None, offsets [T] float64 added to the device plane)."""
        b = self.spec.buildings[i]
        T = tab.n_steps
        w = slice(tab.start, tab.end + 1)
        ts = tab.ts[:, i].astype(np.float64)
        zeros = np.zeros(T)
        dyn = b.is_dynamics and b.dynamics is not None
        cc = charging_constraint_observations(b)
        if k in cc:
            # headroom / violation follow the charger actions of the step just simulated -- they are not time series, so the
            # reference returns current values in every observation mode (building.py:1462-1481); reset(): limit / 0
            if k.startswith('charging_phase_one_hot_'):
                return cc[k] * np.ones(T), None, zeros
            fb = list(tab.flex.flex_bldg).index(i)
            names_ = [n for n, _ in b.charging_constraints.headroom_keys()]
            if k == 'charging_constraint_violation_kwh':
                plane = abi.CLX_VIOLATION
            elif k == 'charging_building_headroom_kw':
                plane = abi.CLX_HEADROOM
            else:
                limited = [ph['name'] for ph in b.charging_constraints.phases]
                plane = abi.CLX_HEADROOM_PHASE0 + limited.index(k[len('charging_phase_'):-len('_headroom_kw')])
            self._forced_sources[(i, k)] = (SRC_EXTRA, plane, fb)
            return cc[k] * np.ones(T), None, zeros
        if tab.flex is not None and k in tab.flex.observations:
            # charger / washing-machine observations are functions of the schedule row (flex.py); row 0 is what reset() returns
            values = np.array(tab.flex.observations[k][:T], dtype=np.float64)
            values[0] = tab.flex.reset_observations[k][0]
            return values, None, zeros
        if k == 'solar_generation':
            return np.abs(ts[:, abi.CLT_SOLAR]), None, zeros
        if k == 'power_outage':
            return tab.outage[:, i].astype(np.float64), None, zeros
        if k == 'cooling_device_efficiency':
            return ts[:, abi.CLT_COP_COOL], None, zeros
        if k == 'heating_device_efficiency':
            return ts[:, abi.CLT_COP_HEAT], None, zeros
        if k == 'dhw_device_efficiency':
            return ts[:, abi.CLT_COP_DHW], None, zeros
        if k in ('indoor_dry_bulb_temperature_cooling_delta', 'indoor_dry_bulb_temperature_heating_delta'):
            sp = np.asarray(b.series['indoor_dry_bulb_temperature_cooling_set_point' if 'cooling' in k else
                                     'indoor_dry_bulb_temperature_heating_set_point'][w], dtype=np.float64)
            temp = np.asarray(b.series['indoor_dry_bulb_temperature'][w], dtype=np.float64)
            return temp - sp, ((SRC_TEMP, 0) if dyn else None), -sp
        if k == 'indoor_dry_bulb_temperature':
            return np.asarray(b.series[k][w], dtype=np.float64), ((SRC_TEMP, 0) if dyn else None), zeros
        if k in ('cooling_demand', 'heating_demand', 'dhw_demand'):
            # the delivered-energy series start as copies of the demand series (building.py:2555-2557), so the
            # not-yet-simulated slot the reference reads holds the data-file demand
            return np.asarray(b.series[k][w], dtype=np.float64), _DEVICE_SOURCE[k], zeros
        if k in ENV_DEPENDENT:
            stale = zeros.copy()                 # the reference reads the not-yet-simulated (zero) slot of t+1 (App. B3)
            stale[0] = self._reset_series(i, k, tab)[0]
            return stale, _DEVICE_SOURCE.get(k), zeros
        if k in b.series and isinstance(b.series[k], np.ndarray):
            return np.asarray(b.series[k][w], dtype=np.float64), None, zeros
        raise KeyError(f'observation {k!r} cannot be produced for building {b.name}')

    def _reset_series(self, i: int, k: str, tab: EpisodeTables) -> Optional[np.ndarray]:
        """Value of env-dependent observation `k` right after `reset()` (the reference's `update_variables` at t = 0,
        building.py:2618-2652) for an episode starting at each table row; None for env-independent observations."""
        if tab.flex is not None and k in tab.flex.reset_observations:
            return np.array(tab.flex.reset_observations[k][:tab.n_steps], dtype=np.float64)
        if k not in ENV_DEPENDENT or k in ('cooling_demand', 'heating_demand', 'dhw_demand'):
            return None
        ts = tab.ts[:, i].astype(np.float64)
        pf = tab.params_f32()[i][:abi.CLP_USED].astype(np.float64)        # (the float32 slots; the words behind them hold flags and float64 halves)
        ones = np.ones(tab.n_steps)
        const = {'electrical_storage_soc': pf[abi.CLP_B_SOC0], 'cooling_storage_soc': pf[abi.CLP_CS_SOC0],
                 'heating_storage_soc': pf[abi.CLP_HS_SOC0], 'dhw_storage_soc': pf[abi.CLP_DS_SOC0]}
        if k in const:
            return const[k] * ones
        if not self.reference_quirks:
            return 0.0 * ones
        heat_hp = bool(tab.params[i, abi.CLP_FLAGS] & abi.CLF_HEAT_IS_HP)
        r = float(self.spec.buildings[i].time_step_ratio)      # Device.electricity_consumption = accumulator * ratio (energy_model.py:118)
        c_cool = ts[:, abi.CLT_COOL_DEM] * ts[:, abi.CLT_ICOP_COOL] * r
        c_heat = ts[:, abi.CLT_HEAT_DEM] * (ts[:, abi.CLT_ICOP_HEAT] if heat_hp else pf[abi.CLP_T0_IHEAT_DIV]) * r
        c_dhw = ts[:, abi.CLT_DHW_DEM] * ts[:, abi.CLT_ICOP_DHW] * r
        net = np.where(ts[:, abi.CLT_OUTAGE] != 0, 0.0, c_cool + c_heat + c_dhw + ts[:, abi.CLT_NSL] * r + ts[:, abi.CLT_SOLAR])
        return {'cooling_electricity_consumption': c_cool, 'heating_electricity_consumption': c_heat,
                'dhw_electricity_consumption': c_dhw, 'net_electricity_consumption': net}.get(k, 0.0 * ones)

    def episode(self, tab: EpisodeTables, reset_table: bool = False) -> 'ObservationTables':
        """Pack the episode window: host table (row r = observation returned when ``time_step == r``) and device map.
        `reset_table`: also pack, for every table row r, the observation `reset()` returns for an episode that STARTS at
        row r (per-env-block episode windows, `cl_dims.env_row0`): rows >= 1 of `table` are start-independent, row 0 is not."""
        T, N = tab.n_steps, self.n_cols
        self._forced_sources: Dict[Tuple[int, str], Tuple[int, int, int]] = {}
        table = np.zeros((T, N), dtype=np.float64)
        resets = np.zeros((T, N), dtype=np.float64) if reset_table else None
        src = np.full(N, -1, dtype=np.int32)
        scale = np.zeros(N, dtype=np.float64)
        needs_detail = False
        unsupported: List[str] = []
        lo, hi = self.limits()
        for c, (i, name) in enumerate(self.columns):
            raw, part = name, None
            if self.normalize and name.rsplit('_', 1)[0] in PERIODIC and name.rsplit('_', 1)[-1] in ('cos', 'sin'):
                raw, part = name.rsplit('_', 1)
            values, source, offset = self._raw_column(i, raw, tab)
            # CLO_B_EB holds the battery's energy balance; the observation is its electricity consumption, i.e. the
            # balance times time_step_ratio (energy_model.py:118)
            gain = float(self.spec.buildings[i].time_step_ratio) if raw == 'electrical_storage_electricity_consumption' else 1.0
            if part is not None:
                x = 2 * np.pi * values / PERIODIC[raw]
                values = np.cos(x) if part == 'cos' else np.sin(x)
            a, b0 = 1.0, 0.0                                   # normalisation obs' = a * obs + b0
            if self.normalize:
                if lo[c] == hi[c]:
                    a, b0 = 0.0, 0.0                           # preprocessing.py:143-144
                else:
                    a, b0 = 1.0 / (hi[c] - lo[c]), -lo[c] / (hi[c] - lo[c])
            if resets is not None:
                rv = self._reset_series(i, raw, tab)
                resets[:, c] = a * (values if rv is None else rv) + b0
            if (i, raw) in self._forced_sources:
                # current value of the step just simulated in every mode: obs = plane * a + b0 (row 0: the reset value)
                kind, plane, fb = self._forced_sources[(i, raw)]
                src[c] = (kind << 28) | (plane << 20) | fb
                scale[c] = a
                table[1:, c] = b0
                table[0, c] = a * values[0] + b0
            elif self.mode == 'current' and source is not None:
                # row r (r >= 1) pairs exogenous values of r with env-dependent values computed at r - 1
                kind, plane = source
                if kind == SRC_OUT and plane in _DETAIL_PLANES:
                    needs_detail = True
                src[c] = (kind << 28) | (plane << 20) | i
                scale[c] = a * gain
                table[1:, c] = a * offset[:-1] + b0
                table[0, c] = a * values[0] + b0
            else:
                if self.mode == 'current' and raw in _NO_DEVICE_PLANE:
                    unsupported.append(raw)
                table[:, c] = a * values + b0
        if unsupported:
            raise NotImplementedError(f"observation_mode='current' has no device plane for {sorted(set(unsupported))}")
        return ObservationTables(table=table, col_src=src, col_scale=scale.astype(np.float32), needs_detail=needs_detail,
                                 reset_table=resets)


class ObservationTables:
    def __init__(self, table: np.ndarray, col_src: np.ndarray, col_scale: np.ndarray, needs_detail: bool,
                 reset_table: Optional[np.ndarray] = None):
        self.table, self.col_src, self.col_scale, self.needs_detail = table, col_src, col_scale, needs_detail
        self.reset_table = reset_table

    @property
    def n_dependent(self) -> int:
        return int((self.col_src >= 0).sum())

    def compact(self):
        """``(tables of the env-dependent columns only, their column indices)``: the factorised observation -- one shared row
        for the whole batch + an ``[n_env, n_dependent]`` matrix -- that `VectorCityLearnEnv(observations='compact')` returns
        instead of materialising ``n_env`` copies of the ~93 % of columns that do not depend on the env."""
        cols = np.nonzero(self.col_src >= 0)[0]
        sub = ObservationTables(np.ascontiguousarray(self.table[:, cols]), self.col_src[cols].copy(), self.col_scale[cols].copy(), self.needs_detail,
                                None if self.reset_table is None else np.ascontiguousarray(self.reset_table[:, cols]))
        return sub, cols

    def host_row(self, r: int, state: Optional[np.ndarray] = None, out_bldg: Optional[np.ndarray] = None,
                 indoor_temp: Optional[np.ndarray] = None, extra: Optional[np.ndarray] = None) -> np.ndarray:
        """Observation vector of ONE environment at row `r` computed on the host from host copies of the device
        planes (`state [CL_NS, B]`, `out_bldg [CL_NO, B]`, `indoor_temp [B]`): what `cl_observe_f32` writes."""
        row = self.table[r].copy()
        if r == 0:
            return row
        for c in np.nonzero(self.col_src >= 0)[0]:
            s = int(self.col_src[c])
            kind, plane, b = s >> 28, (s >> 20) & 0xFF, s & 0xFFFFF
            x = state[plane, b] if kind == SRC_STATE else out_bldg[plane, b] if kind == SRC_OUT else \
                indoor_temp[b] if kind == SRC_TEMP else extra[plane, b]
            row[c] = float(x) * float(self.col_scale[c]) + row[c]
        return row
