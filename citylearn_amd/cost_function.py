"""KPI library (mirror of the reference's ``citylearn/cost_function.py``, numpy instead of pandas).

Every function returns the *rolling* series like the reference (``CityLearnEnv.evaluate`` takes the last
element, citylearn.py:1221-1298); results agree with the pandas implementation to float64 round-off.
Citations: /root/reference/citylearn/cost_function.py.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple, Union

import numpy as np

DEFAULT_COMFORT_BAND = 2.0


def _a(x) -> np.ndarray:
    return np.asarray(x, dtype=np.float64)


def _group_reduce(x: np.ndarray, window: int):
    """mean and max over consecutive groups of `window` samples (``data.index / window`` truncated)."""
    n = len(x)
    starts = np.arange(0, n, window)
    mean = np.array([np.nanmean(x[s:s + window]) if np.any(~np.isnan(x[s:s + window])) else np.nan for s in starts])
    mx = np.array([np.nanmax(x[s:s + window]) if np.any(~np.isnan(x[s:s + window])) else np.nan for s in starts])
    return mean, mx


def _expanding_mean(x: np.ndarray) -> np.ndarray:
    """`rolling(window=len, min_periods=1).mean()` skipping NaN."""
    valid = ~np.isnan(x)
    csum = np.cumsum(np.where(valid, x, 0.0))
    cnt = np.cumsum(valid)
    with np.errstate(invalid='ignore', divide='ignore'):
        return np.where(cnt > 0, csum / cnt, np.nan)


class CostFunction:
    @staticmethod
    def ramping(net_electricity_consumption: Sequence[float], down_ramp: bool = None, net_export: bool = None) -> List[float]:
        """Rolling sum of |E_i - E_{i-1}| (or only up-ramps by default); cost_function.py:10-59."""
        down_ramp = False if down_ramp is None else down_ramp
        net_export = True if net_export is None else net_export
        e = _a(net_electricity_consumption)
        r = np.full(len(e), np.nan)
        r[1:] = e[1:] - e[:-1]
        r = np.abs(r) if down_ramp else np.where(np.isnan(r), np.nan, np.maximum(r, 0.0))
        if not net_export:
            r[e < 0] = 0.0
        return np.cumsum(np.where(np.isnan(r), 0.0, r)).tolist()

    @staticmethod
    def one_minus_load_factor(net_electricity_consumption: Sequence[float], window: int = None) -> List[float]:
        """Rolling mean over groups of `window` steps of 1 - mean/max; cost_function.py:62-86."""
        window = 730 if window is None else window
        mean, mx = _group_reduce(_a(net_electricity_consumption), window)
        with np.errstate(invalid='ignore', divide='ignore'):
            lf = 1 - mean / mx
        return _expanding_mean(lf).tolist()

    @staticmethod
    def peak(net_electricity_consumption: Sequence[float], window: int = None) -> List[float]:
        """Rolling mean of the per-group maxima; cost_function.py:89-111."""
        window = 24 if window is None else window
        _, mx = _group_reduce(_a(net_electricity_consumption), window)
        return _expanding_mean(mx).tolist()

    @staticmethod
    def electricity_consumption(net_electricity_consumption: Sequence[float]) -> List[float]:
        return np.cumsum(np.clip(_a(net_electricity_consumption), 0, None)).tolist()      # cost_function.py:114-133

    @staticmethod
    def zero_net_energy(net_electricity_consumption: Sequence[float]) -> List[float]:
        return np.cumsum(_a(net_electricity_consumption)).tolist()                          # cost_function.py:136-156

    @staticmethod
    def carbon_emissions(carbon_emissions: Sequence[float]) -> List[float]:
        return np.cumsum(np.clip(_a(carbon_emissions), 0, None)).tolist()                   # cost_function.py:159-176

    @staticmethod
    def cost(cost: Sequence[float]) -> List[float]:
        return np.cumsum(np.clip(_a(cost), 0, None)).tolist()                               # cost_function.py:179-196

    @staticmethod
    def quadratic(net_electricity_consumption: Sequence[float]) -> List[float]:
        return np.cumsum(np.clip(_a(net_electricity_consumption), 0, None) ** 2).tolist()   # cost_function.py:199-221

    @staticmethod
    def discomfort(indoor_dry_bulb_temperature, dry_bulb_temperature_cooling_set_point, dry_bulb_temperature_heating_set_point,
                   band: Union[float, Sequence[float]] = None, occupant_count: Sequence[int] = None) -> Tuple[list, ...]:
        """Rolling comfort statistics; cost_function.py:224-321."""
        temp = _a(indoor_dry_bulb_temperature)
        n = len(temp)
        occ = np.ones(n) if occupant_count is None else _a(occupant_count)
        band = np.full(n, DEFAULT_COMFORT_BAND) if band is None else np.broadcast_to(_a(band), (n,)).copy()
        occupied = int((occ > 0.0).sum())
        cd = temp - _a(dry_bulb_temperature_cooling_set_point)
        hd = temp - _a(dry_bulb_temperature_heating_set_point)
        cd[occ == 0.0] = 0.0
        hd[occ == 0.0] = 0.0
        hot = cd > band
        cold = hd < -band
        with np.errstate(invalid='ignore', divide='ignore'):
            unmet = np.cumsum((hot | cold).astype(float)) / occupied
            cold_p = np.cumsum(cold.astype(float)) / occupied
            hot_p = np.cumsum(hot.astype(float)) / occupied
        cmag = np.abs(np.minimum(hd, 0.0))
        hmag = np.abs(np.maximum(cd, 0.0))

        def _roll(x, fn):
            valid = ~np.isnan(x)
            out = np.full(n, np.nan)
            if fn == 'min':
                acc = np.fmin.accumulate(np.where(valid, x, np.inf))
                out = np.where(np.cumsum(valid) > 0, acc, np.nan)
            elif fn == 'max':
                acc = np.fmax.accumulate(np.where(valid, x, -np.inf))
                out = np.where(np.cumsum(valid) > 0, acc, np.nan)
            else:
                out = _expanding_mean(x)
            return out.tolist()

        return (unmet.tolist(), cold_p.tolist(), hot_p.tolist(), _roll(cmag, 'min'), _roll(cmag, 'max'), _roll(cmag, 'mean'),
                _roll(hmag, 'min'), _roll(hmag, 'max'), _roll(hmag, 'mean'))

    @staticmethod
    def one_minus_thermal_resilience(power_outage: Sequence[int], **kwargs):
        """Discomfort proportion restricted to outage steps; cost_function.py:324-353."""
        occ = kwargs.get('occupant_count')
        occ = np.ones(len(power_outage), dtype='float32') if occ is None else np.array(occ, dtype='float32')
        occ[np.array(power_outage, dtype='float32') == 0.0] = 0.0
        kwargs['occupant_count'] = occ
        return CostFunction.discomfort(**kwargs)[0]

    @staticmethod
    def normalized_unserved_energy(expected_energy: Sequence[float], served_energy: Sequence[float],
                                   power_outage: Sequence[int] = None) -> List[float]:
        """Rolling unserved / total expected energy (outage steps only if a signal is given); cost_function.py:356-388."""
        exp, srv = _a(expected_energy).copy(), _a(served_energy)
        po = np.ones(len(srv)) if power_outage is None else _a(power_outage)
        un = exp - srv
        un[po == 0] = 0.0
        exp[po == 0] = 0.0
        with np.errstate(invalid='ignore', divide='ignore'):
            return (np.cumsum(un) / exp.sum()).tolist()
