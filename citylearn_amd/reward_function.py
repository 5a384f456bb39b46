"""Reward-function plugin surface (mirror of the reference's ``citylearn/reward_function.py``).

Same constructor / ``env_metadata`` / ``reset()`` / ``calculate(observations)`` contract as the reference
(reward_function.py:7-88; plugin how-to in examples/custom_reward_function.py), so user subclasses written for
the reference keep working: they are called on the host with per-building observation dictionaries built from
the device outputs (``CityLearnEnv`` host path).  The stock classes below additionally carry ``device_kind``:
when the env's reward function is *exactly* one of them, the reward is computed inside the HIP step kernel
(``cl_unit.h: unit_reward / marl_reward``) and ``calculate`` is never called on the hot path.
"""
from __future__ import annotations

from typing import Any, List, Mapping, Optional, Tuple, Union

import numpy as np

from . import abi

ZERO_DIVISION_PLACEHOLDER = 1e-6


class RewardFunction:
    """Default reward: ``-max(net_electricity_consumption, 0) ** exponent`` per building
    (reference reward_function.py:65-88); summed into one value when ``central_agent``."""

    device_kind: Optional[int] = abi.CLR_DEFAULT

    def __init__(self, env_metadata: Mapping[str, Any], exponent: float = None, **kwargs):
        coefficient = kwargs.pop('charging_constraint_penalty_coefficient', None)     # reference reward_function.py:22-25, 49-58
        self.charging_constraint_penalty_coefficient = 1.0 if coefficient is None else float(coefficient)
        self.env_metadata = env_metadata
        self.exponent = 1.0 if exponent is None else exponent

    @property
    def env_metadata(self) -> Mapping[str, Any]:
        return self._env_metadata

    @env_metadata.setter
    def env_metadata(self, env_metadata: Mapping[str, Any]):
        self._env_metadata = env_metadata

    @property
    def central_agent(self) -> bool:
        return self.env_metadata['central_agent']

    def reset(self):
        pass

    def _finish(self, per_building: List[float]) -> List[float]:
        return [sum(per_building)] if self.central_agent else list(per_building)

    def calculate(self, observations: List[Mapping[str, Union[int, float]]]) -> List[float]:
        return self._finish([-(max(o['net_electricity_consumption'], 0) ** self.exponent) for o in observations])


class MARL(RewardFunction):
    """``sign(-e) * 0.01 * e**2 * max(0, district e)`` (reference reward_function.py:120-143)."""

    device_kind = abi.CLR_MARL

    def __init__(self, env_metadata: Mapping[str, Any]):
        super().__init__(env_metadata)

    def calculate(self, observations):
        e = np.array([o['net_electricity_consumption'] for o in observations], dtype=float)
        district = max(0.0, float(e.sum()))
        r = np.sign(-e) * 0.01 * e ** 2 * district
        return [float(r.sum())] if self.central_agent else r.tolist()


class IndependentSACReward(RewardFunction):
    """``min(-e, 0)`` (reference reward_function.py:145-168; its ``v*-1**3`` parses as ``-v``)."""

    device_kind = abi.CLR_INDEPENDENT_SAC

    def __init__(self, env_metadata: Mapping[str, Any]):
        super().__init__(env_metadata)

    def calculate(self, observations):
        return self._finish([min(-o['net_electricity_consumption'], 0) for o in observations])


class SolarPenaltyReward(RewardFunction):
    """``sum over storages with capacity > 1e-6 of -(1 + sign(e) * soc) * |e|`` (reference reward_function.py:170-214)."""

    device_kind = abi.CLR_SOLAR_PENALTY

    def __init__(self, env_metadata: Mapping[str, Any]):
        super().__init__(env_metadata)

    def calculate(self, observations):
        out = []
        for o, m in zip(observations, self.env_metadata['buildings']):
            e = o['net_electricity_consumption']
            r = 0.0
            for key in ('cooling_storage', 'heating_storage', 'dhw_storage', 'electrical_storage'):
                if m[key]['capacity'] > ZERO_DIVISION_PLACEHOLDER:
                    r += -(1.0 + np.sign(e) * o.get(f'{key}_soc', 0.0)) * abs(e)
            out.append(r)
        return self._finish(out)


class ComfortReward(RewardFunction):
    """Thermal-comfort reward (reference reward_function.py:216-334).  Fused into the LSTM indoor-temperature
    kernel (`cl_lstm_step_f32`, csrc/cl_lstm.h: comfort_reward) when the env's reward function is exactly this class."""

    device_kind = 'comfort'

    def __init__(self, env_metadata: Mapping[str, Any], band: float = None, lower_exponent: float = None,
                 higher_exponent: float = None):
        super().__init__(env_metadata)
        self.band = band
        self.lower_exponent = 2.0 if lower_exponent is None else lower_exponent
        self.higher_exponent = 2.0 if higher_exponent is None else higher_exponent

    def _one(self, o: Mapping[str, float]) -> float:
        heating = o.get('heating_demand', 0.0) > o.get('cooling_demand', 0.0)
        mode = o['hvac_mode']
        temp = o['indoor_dry_bulb_temperature']
        band = self.band if self.band is not None else o['comfort_band']
        if mode in (1, 2):
            sp = o['indoor_dry_bulb_temperature_cooling_set_point'] if mode == 1 else o['indoor_dry_bulb_temperature_heating_set_point']
            lo, hi, delta = sp - band, sp + band, abs(temp - sp)
            if temp < lo:
                return -(delta ** (self.lower_exponent if mode == 2 else self.higher_exponent))
            if temp < sp:
                return 0.0 if heating else -delta
            if temp <= hi:
                return -delta if heating else 0.0
            return -(delta ** (self.higher_exponent if heating else self.lower_exponent))
        csp, hsp = o['indoor_dry_bulb_temperature_cooling_set_point'], o['indoor_dry_bulb_temperature_heating_set_point']
        lo, hi = hsp - band, csp + band
        cd, hd = temp - csp, temp - hsp
        if temp < lo:
            return -(abs(hd) ** (self.higher_exponent if not heating else self.lower_exponent))
        if temp < hsp:
            return -abs(hd)
        if temp <= csp:
            return 0.0
        if temp < hi:
            return -abs(cd)
        return -(abs(cd) ** (self.higher_exponent if heating else self.lower_exponent))

    def calculate(self, observations):
        return self._finish([self._one(o) for o in observations])


class SolarPenaltyAndComfortReward(RewardFunction):
    """Weighted sum of :class:`SolarPenaltyReward` and :class:`ComfortReward` (reference reward_function.py:336-386)."""

    device_kind = None

    def __init__(self, env_metadata: Mapping[str, Any], band: float = None, lower_exponent: float = None,
                 higher_exponent: float = None, coefficients: Tuple = None):
        self._functions = [SolarPenaltyReward(env_metadata),
                           ComfortReward(env_metadata, band=band, lower_exponent=lower_exponent, higher_exponent=higher_exponent)]
        super().__init__(env_metadata)
        self.coefficients = [1.0] * 2 if coefficients is None else list(coefficients)
        assert len(self.coefficients) == 2, 'SolarPenaltyAndComfortReward needs 2 coefficients.'

    @RewardFunction.env_metadata.setter
    def env_metadata(self, env_metadata):
        RewardFunction.env_metadata.fset(self, env_metadata)
        for f in self._functions:
            f.env_metadata = env_metadata

    def calculate(self, observations):
        r = np.array([f.calculate(observations) for f in self._functions], dtype='float32')
        return (r * np.reshape(self.coefficients, (2, 1))).sum(axis=0).tolist()


class Electric_Vehicles_Reward_Function(MARL):
    """EV-charging reward on top of :class:`MARL` (reference reward_function.py:389-531): per building with chargers,
    the sum over its *connected* chargers of the weighted terms below, scaled by ``1 / (1 + |MARL reward|)``; buildings
    without chargers get 0.  (The 'no_car_charging' term is computed for unconnected chargers and then skipped by the
    reference's ``continue`` -- it never reaches the total.)  On the device: `cl_flex_kernel` + the step kernel's
    district sweep (csrc/cl_flex.h, cl_unit.h: ev_reward) when the env's reward function is exactly this class."""

    device_kind = abi.CLR_EV

    def __init__(self, env_metadata: Mapping[str, Any], weights: Mapping[str, float] = None):
        super().__init__(env_metadata)
        self.weights = weights or {
            'no_car_charging': -5.0, 'battery_limits': -2.0, 'soc_impossible': -10.0, 'soc_under': -5.0,
            'close_soc': 10.0, 'self_ev_consumption': 5.0, 'extra_self_production': 5.0,
        }

    def calculate(self, observations):
        current = MARL.calculate(self, observations)
        out = []
        for i, o in enumerate(observations):
            info = o.get('electric_vehicles_chargers_dict', {})
            if not info:
                reward = 0
            else:
                reward = self.calculate_ev_penalty(o, current[0] if self.central_agent else current[i])
            violation = float(o.get('charging_constraint_violation_kwh', 0.0) or 0.0)
            reward -= violation * self.charging_constraint_penalty_coefficient if violation > 0.0 else 0.0
            out.append(reward)
        return [sum(out)] if self.central_agent else out

    def calculate_ev_penalty(self, o: Mapping[str, Any], current_reward: float) -> float:
        w = self.weights
        net = o.get('net_electricity_consumption', 0)
        mult = 1.0 / (1.0 + abs(current_reward))
        total = 0.0
        for data in o.get('electric_vehicles_chargers_dict', {}).values():
            if not data['connected']:
                continue
            capacity, last = data['battery_capacity'], data.get('last_charged_kwh') or 0.0
            hours = data.get('hours_until_departure', 0)
            k = 0.0
            held = data['previous_battery_soc'] * capacity + last
            if held > capacity or held < data['min_capacity']:
                k += w['battery_limits'] * mult
            if data.get('required_soc') is not None:
                diff = data['battery_soc'] - data['required_soc']
                diff_kwh = diff * capacity
                reach_c, reach_d = data.get('max_charging_power', 0) * hours, data.get('max_discharging_power', 0) * hours
                if diff_kwh > reach_c:
                    k += w['soc_impossible'] * mult
                if hours == 0:
                    if -0.25 < diff <= -0.10:
                        k += 2 * w['soc_under'] * mult
                    elif diff <= -0.25:
                        k += (w['soc_under'] ** 2) * mult
                    elif -0.10 < diff <= 0.10:
                        k += w['close_soc'] * mult
                if abs(diff_kwh) <= max(reach_c, reach_d):
                    k += w['close_soc'] * mult * (1.0 / (hours + 0.1))
            if last > 0 and net < 0:
                k += w['extra_self_production'] * mult
            elif last < 0 and net < 0:
                k += -0.5 * w['extra_self_production'] * mult
            if last < 0 and net > 0:
                k += w['self_ev_consumption'] * mult
            elif last > 0 and net > 0:
                k += -0.5 * w['self_ev_consumption'] * mult
            total += k
        return total


BUILTIN = {c.__name__: c for c in (RewardFunction, MARL, IndependentSACReward, SolarPenaltyReward, ComfortReward,
                                    SolarPenaltyAndComfortReward, Electric_Vehicles_Reward_Function)}


def resolve(reward_type: Union[str, type, None]):
    """Map the schema's dotted ``reward_function.type`` (or a class / dotted path given as kwarg, reference
    citylearn.py:2143-2163) to a class.  ``citylearn.reward_function.X`` resolves to the stock class ``X`` here;
    anything else is imported by its dotted path (user plugins)."""
    if reward_type is None:
        return RewardFunction
    if isinstance(reward_type, type):
        return reward_type
    module_name, _, class_name = reward_type.rpartition('.')
    if module_name in ('citylearn.reward_function', 'citylearn_amd.reward_function', '') and class_name in BUILTIN:
        return BUILTIN[class_name]
    import importlib
    return getattr(importlib.import_module(module_name), class_name)
