"""`evaluate()` KPIs (reference citylearn.py:1136-1323) from per-step series, independent of where they came from."""
from __future__ import annotations

import numpy as np

from . import abi
from .cost_function import CostFunction


def evaluate_district(spec, tables, K: int, net, base, cost, emission, expected, served, d_net, comfort_band: float = None):
    """`net`, `base`, `cost`, `emission`, `expected`, `served`: float arrays ``[K, n_bldg]`` for the K completed steps
    (control net, baseline net, control cost / emission, expected / served energy); `d_net`: the K district sums.
    Returns the reference's ``DataFrame[cost_function, value, name, level]``."""
    import pandas as pd
    net, base, cost, emission = (np.asarray(a, dtype='float32') for a in (net, base, cost, emission))
    expected, served = np.asarray(expected, dtype='float32').copy(), np.asarray(served, dtype='float32').copy()
    comfort_band = 2.0 if comfort_band is None else comfort_band
    B = len(spec.buildings)
    tab = tables

    def safe_div(c, b):
        c = float(c) if np.isfinite(c) else 0.0
        b = float(b) if np.isfinite(b) else 0.0
        if b == 0.0:
            return 1.0 if c == 0.0 else None
        return c / b

    # building series have time_step + 1 entries; the last one is the untouched (zero) slot of step K
    pad = lambda a: np.concatenate([a, np.zeros((1, B), dtype=a.dtype)], axis=0) if K < tab.n_steps else a
    net, base = pad(net), pad(base)
    cost_c, em_c = pad(cost), pad(emission)
    expected, served = pad(expected), pad(served)
    n = net.shape[0]
    price = tab.ts[:n, :, abi.CLT_PRICE].astype(np.float64)
    carbon = tab.ts[:n, :, abi.CLT_CARBON].astype(np.float64)
    if K < tab.n_steps:
        # the (K+1)-th entry of the demand-type series is the dataset value, not zero
        w = tab.start + K
        for i, b in enumerate(spec.buildings):
            tot = float(b.series['cooling_demand'][w]) + float(b.series['heating_demand'][w]) + float(b.series['dhw_demand'][w]) \
                + float(b.series['non_shiftable_load'][w])
            expected[-1, i] = tot
            served[-1, i] = tot
    rows = []
    for i, b in enumerate(spec.buildings):
        w = slice(tab.start, tab.start + n)
        base_cost = price[:, i] * base[:, i]
        base_em = np.clip(carbon[:, i] * base[:, i], 0, None)
        vals = {
            'electricity_consumption_total': safe_div(CostFunction.electricity_consumption(net[:, i])[-1], CostFunction.electricity_consumption(base[:, i])[-1]),
            'zero_net_energy': safe_div(CostFunction.zero_net_energy(net[:, i])[-1], CostFunction.zero_net_energy(base[:, i])[-1]),
            'carbon_emissions_total': safe_div(CostFunction.carbon_emissions(em_c[:, i])[-1],
                                               CostFunction.carbon_emissions(base_em)[-1] if float(np.sum(b.series['carbon_intensity'][tab.start:tab.end + 1])) != 0 else 0),
            'cost_total': safe_div(CostFunction.cost(cost_c[:, i])[-1],
                                   CostFunction.cost(base_cost)[-1] if float(np.sum(b.series['electricity_pricing'][tab.start:tab.end + 1])) != 0 else 0),
        }
        kw = dict(indoor_dry_bulb_temperature=b.series['indoor_dry_bulb_temperature'][w],
                  dry_bulb_temperature_cooling_set_point=b.series['indoor_dry_bulb_temperature_cooling_set_point'][w],
                  dry_bulb_temperature_heating_set_point=b.series['indoor_dry_bulb_temperature_heating_set_point'][w],
                  band=comfort_band, occupant_count=b.series['occupant_count'][w])
        d = CostFunction.discomfort(**kw)
        for name, series in zip(('discomfort_proportion', 'discomfort_cold_proportion', 'discomfort_hot_proportion',
                                 'discomfort_cold_delta_minimum', 'discomfort_cold_delta_maximum', 'discomfort_cold_delta_average',
                                 'discomfort_hot_delta_minimum', 'discomfort_hot_delta_maximum', 'discomfort_hot_delta_average'), d):
            vals[name] = series[-1]
        po = tab.outage[:n, i]
        vals['one_minus_thermal_resilience_proportion'] = CostFunction.one_minus_thermal_resilience(power_outage=po, **kw)[-1]
        vals['power_outage_normalized_unserved_energy_total'] = CostFunction.normalized_unserved_energy(expected[:, i], served[:, i], power_outage=po)[-1]
        vals['annual_normalized_unserved_energy_total'] = CostFunction.normalized_unserved_energy(expected[:, i], served[:, i])[-1]
        for k, v in vals.items():
            rows.append({'cost_function': k, 'value': v, 'name': b.name, 'level': 'building'})
    building_level = pd.DataFrame(rows)
    d_c = np.asarray(d_net, dtype=np.float64)                    # K entries (citylearn.py:1909-1918)
    d_b = base.astype(np.float64).sum(axis=1)                               # K + 1 entries (sum of building series)
    district = {
        'ramping_average': safe_div(CostFunction.ramping(d_c)[-1], CostFunction.ramping(d_b)[-1]),
        'daily_one_minus_load_factor_average': safe_div(CostFunction.one_minus_load_factor(d_c, window=24)[-1],
                                                        CostFunction.one_minus_load_factor(d_b, window=24)[-1]),
        'monthly_one_minus_load_factor_average': safe_div(CostFunction.one_minus_load_factor(d_c, window=730)[-1],
                                                          CostFunction.one_minus_load_factor(d_b, window=730)[-1]),
        'daily_peak_average': safe_div(CostFunction.peak(d_c, window=24)[-1], CostFunction.peak(d_b, window=24)[-1]),
        'all_time_peak_average': safe_div(CostFunction.peak(d_c, window=tab.n_steps)[-1], CostFunction.peak(d_b, window=tab.n_steps)[-1]),
    }
    district_level = pd.DataFrame([{'cost_function': k, 'value': v} for k, v in district.items()])
    district_level = pd.concat([district_level, building_level[['cost_function', 'value']]], ignore_index=True, sort=False)
    district_level['value'] = pd.to_numeric(district_level['value'], errors='coerce')
    district_level = district_level.groupby(['cost_function'])[['value']].mean().reset_index()
    district_level['name'] = 'District'
    district_level['level'] = 'district'
    return pd.concat([district_level, building_level], ignore_index=True, sort=False)

