"""CPU oracle of the adjacent LSTM indoor-temperature stage (TEST INFRASTRUCTURE, NOT PRODUCT CODE).

Restates `LSTMDynamicsBuilding._update_dynamics_input / get_dynamics_input / update_indoor_dry_bulb_temperature`
(/root/reference/citylearn/building.py:3000-3078) and `LSTMDynamics.forward` (citylearn/dynamics.py:50-127) for ONE
building of ONE environment, with `torch.nn.LSTM` on the CPU exactly like the reference.  Pinned against the
reference's own predicted temperatures (tests/golden/g2023_p2/reference.npz `indoor_temp`) in
tests/test_oracle_golden.py::test_lstm_oracle_matches_reference.

Also holds a scalar restatement of `ComfortReward.calculate` (citylearn/reward_function.py:269-334).
"""
from __future__ import annotations

import numpy as np
import torch


class LSTMOracle:
    def __init__(self, bspec, tables, b_index: int):
        d = bspec.dynamics
        self.spec, self.d = bspec, d
        self.names = list(d.input_observation_names)
        self.lo = list(d.input_normalization_minimum)
        self.hi = list(d.input_normalization_maximum)
        self.lstm = torch.nn.LSTM(input_size=d.input_size, hidden_size=d.hidden_size, num_layers=d.num_layers, batch_first=True)
        self.linear = torch.nn.Linear(d.hidden_size, 1)
        sd = torch.load(d.filepath, map_location='cpu')
        sd = sd.get('model_state_dict', sd)
        self.lstm.load_state_dict({k.replace('l_lstm.', ''): v for k, v in sd.items() if k.startswith('l_lstm.')})
        self.linear.load_state_dict({k.replace('l_linear.', ''): v for k, v in sd.items() if k.startswith('l_linear.')})
        w = slice(tables.start, tables.end + 1)
        self.series = {k: v[w] for k, v in bspec.series.items() if isinstance(v, np.ndarray)}
        self.reset()

    def reset(self):                                     # dynamics.py:112-127
        self.hidden = (torch.zeros(self.d.num_layers, 1, self.d.hidden_size), torch.zeros(self.d.num_layers, 1, self.d.hidden_size))
        self.window = [[None] * (self.d.lookback + 1) for _ in self.names]

    def _observation(self, k: str, t: int, delivered_cooling: float, delivered_heating: float = 0.0):
        """`Building.observations(include_all=True, periodic_normalization=True)[k]` at step t (building.py:1115-1219)."""
        for base, x_max in (('month', 12), ('hour', 24), ('day_type', 7)):      # building.py:1493-1498
            if k in (f'{base}_sin', f'{base}_cos'):
                x = 2 * np.pi * self.series[base][t] / x_max             # preprocessing.py:68-72
                return np.sin(x) if k.endswith('_sin') else np.cos(x)
        if k == 'cooling_demand':
            return delivered_cooling                                    # building.py:1435
        if k == 'heating_demand':
            return delivered_heating                                    # building.py:1436
        return self.series[k][t]

    def step(self, t: int, delivered_cooling: float, delivered_heating: float = 0.0):
        """Returns the indoor dry-bulb temperature the reference holds for step t after `apply_actions`."""
        # _update_dynamics_input (building.py:3057-3078)
        obs = [self._observation(k, t, delivered_cooling, delivered_heating) for k in self.names]
        self.window = [l[-self.d.lookback:] + [(o - mn) / (mx - mn)] for l, o, mn, mx in zip(self.window, obs, self.lo, self.hi)]
        ix = self.names.index('indoor_dry_bulb_temperature')
        if self.window[0][0] is None:                                   # simulate_dynamics (building.py:2996-2999)
            return float(self.series['indoor_dry_bulb_temperature'][t])
        # get_dynamics_input (building.py:3039-3055)
        rows = [self.window[i][:-1] if k == 'indoor_dry_bulb_temperature' else self.window[i][1:] for i, k in enumerate(self.names)]
        x = torch.tensor(np.array(rows, dtype='float32').T)[np.newaxis, :, :]
        with torch.no_grad():
            out, self.hidden = self.lstm(x.float(), tuple(h.data for h in self.hidden))
            y = self.linear(out[:, -1, :])
        self.window[ix][-1] = y.item()                                  # building.py:3027-3028
        return (y * (self.hi[ix] - self.lo[ix]) + self.lo[ix]).item()   # building.py:3031-3037


def comfort_reward(o: dict, band, lower_exponent: float, higher_exponent: float) -> float:
    """`ComfortReward.calculate` for one building (reward_function.py:269-334)."""
    heating = o.get('heating_demand', 0.0) > o.get('cooling_demand', 0.0)
    mode, temp = o['hvac_mode'], o['indoor_dry_bulb_temperature']
    band = band if band is not None else o['comfort_band']
    if mode in (1, 2):
        sp = o['indoor_dry_bulb_temperature_cooling_set_point'] if mode == 1 else o['indoor_dry_bulb_temperature_heating_set_point']
        lo, hi, delta = sp - band, sp + band, abs(temp - sp)
        if temp < lo:
            return -(delta ** (lower_exponent if mode == 2 else higher_exponent))
        if lo <= temp < sp:
            return 0.0 if heating else -delta
        if sp <= temp <= hi:
            return -delta if heating else 0.0
        return -(delta ** (higher_exponent if heating else lower_exponent))
    csp, hsp = o['indoor_dry_bulb_temperature_cooling_set_point'], o['indoor_dry_bulb_temperature_heating_set_point']
    lo, hi = hsp - band, csp + band
    cd, hd = temp - csp, temp - hsp
    if temp < lo:
        return -(abs(hd) ** (higher_exponent if not heating else lower_exponent))
    if lo <= temp < hsp:
        return -abs(hd)
    if hsp <= temp <= csp:
        return 0.0
    if csp < temp < hi:
        return -abs(cd)
    return -(abs(cd) ** (higher_exponent if heating else lower_exponent))
