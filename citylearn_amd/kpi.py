"""`evaluate()` KPIs (reference citylearn.py:1136-1323) from per-step series, independent of where they came from."""
from __future__ import annotations

import numpy as np

from . import abi
from .cost_function import CostFunction


def evaluate_district(spec, tables, K: int, net, base, cost, emission, expected, served, d_net, comfort_band: float = None,
                      indoor_temp=None, condition_series=None):
    """`net`, `base`, `cost`, `emission`, `expected`, `served`: float arrays ``[K, n_bldg]`` for the K completed steps
    (control net, baseline net, control cost / emission, expected / served energy); `d_net`: the K district sums.
    `indoor_temp` (optional ``[K, n_bldg]``): simulated indoor temperatures (LSTM stage) for the comfort KPIs; the
    data-file temperatures are used otherwise.  Returns the reference's ``DataFrame[cost_function, value, name, level]``."""
    import pandas as pd
    if condition_series is not None:
        # non-default EvaluationCondition pair: (control, baseline) net series; cost / emission follow from them
        # (building.py:368-411: series * price, max(0, series * carbon)) and so does the district control series
        net, base = (np.asarray(a, dtype='float32') for a in condition_series[:2])
        n0 = net.shape[0]
        price0 = tables.ts[:n0, :, abi.CLT_PRICE].astype(np.float32)
        carbon0 = tables.ts[:n0, :, abi.CLT_CARBON].astype(np.float32)
        cost, emission = net * price0, np.clip(net * carbon0, 0, None)
        if condition_series[2] != '':
            d_net = None        # the env-level property is the sum of the buildings' series, K + 1 entries (citylearn.py:700-760)
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
        temp_series = np.array(b.series['indoor_dry_bulb_temperature'][w], dtype=np.float64)
        if indoor_temp is not None and len(indoor_temp):
            temp_series[:K] = np.asarray(indoor_temp, dtype=np.float64)[:K, i]
        kw = dict(indoor_dry_bulb_temperature=temp_series,
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
    d_c = np.asarray(d_net, dtype=np.float64) if d_net is not None else net.astype(np.float64).sum(axis=1)    # K entries (citylearn.py:1909-1918)
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



# ------------------------------------------------------------------------------------------------------------------
# streaming accumulators (CLD_KPI) -> KPI ratios for every env of a batch
# ------------------------------------------------------------------------------------------------------------------

def _safe_div_t(c, b):
    """`_safe_div` of citylearn.py:1172-1189 on tensors: 0/0 -> 1, x/0 -> NaN."""
    import torch
    out = c / b
    zero_b = b == 0
    out = torch.where(zero_b & (c == 0), torch.ones_like(out), out)
    return torch.where(zero_b & (c != 0), torch.full_like(out, float('nan')), out)


def _series_update(k, t: int, v):
    """Host mirror of `kpi_series_update` (cl_kernels.hip) for one extra sample of a district series."""
    import torch
    P = abi
    if t > 0:
        k[P.CLKE_RAMP] += torch.clamp(v - k[P.CLKE_PREV], min=0)
    k[P.CLKE_PREV] = v
    for window, s_, m_, lf_, n_, peak_ in ((24, P.CLKE_DAY_SUM, P.CLKE_DAY_MAX, P.CLKE_DAY_LF_SUM, P.CLKE_DAY_N, P.CLKE_DAY_PEAK_SUM),
                                            (730, P.CLKE_MON_SUM, P.CLKE_MON_MAX, P.CLKE_MON_LF_SUM, P.CLKE_MON_N, None)):
        if t > 0 and t % window == 0:
            k[lf_] += 1 - (k[s_] / window) / k[m_]
            if peak_ is not None:
                k[peak_] += k[m_]
            k[n_] += 1
            k[s_] = torch.zeros_like(k[s_])
            k[m_] = torch.full_like(k[m_], float('-inf'))
        k[s_] = k[s_] + v
        k[m_] = torch.maximum(k[m_], v)
    k[P.CLKE_ALL_MAX] = torch.maximum(k[P.CLKE_ALL_MAX], v)


def finalize_streaming(kpi_bldg, kpi_env, steps_done: int, episode_rows: int, next_expected=None, next_outage=None,
                       shared_baseline: bool = False):
    """Turn the device accumulators into the KPI ratios of `CityLearnEnv.evaluate` for every env.

    Returns ``(building, district)``: dicts name -> tensor ``[n_bldg, n_env]`` / ``[n_env]``; `district` also holds the
    mean over buildings of every building-level KPI (citylearn.py:1317-1318).  The control district series has
    `steps_done` samples, the baseline one more (the untouched zero slot of the next step, App. A.7) unless the
    data ran out.  `next_expected` / `next_outage` (``[n_bldg]``): expected energy and outage flag of that extra row
    (the reference's series include it in the normalisation, citylearn.py:1216 + cost_function.py:384).
    `shared_baseline` (`StepEngine.kpi_shared_baseline`: battery + PV districts stepped without the detail planes): the baseline sums,
    the expected energy and the baseline district series do not depend on the env there and were kept at the first env of every
    block of ``CL_ROW0_BLOCK`` envs only; they are spread over their blocks here (the slice must start on a block boundary)."""
    import torch
    P = abi
    kb = kpi_bldg.double().clone()
    kpi_env = kpi_env.clone()
    if shared_baseline:
        n_env = kb.shape[-1]
        lead = (torch.arange(n_env, device=kb.device) // P.CL_ROW0_BLOCK) * P.CL_ROW0_BLOCK
        for plane in (P.CLK_B_POS, P.CLK_B_NET, P.CLK_B_EMISSION, P.CLK_B_COST, P.CLK_EXPECTED_ALL):
            kb[plane] = kb[plane][:, lead]
        kpi_env[P.CLKE_PER_COND:] = kpi_env[P.CLKE_PER_COND:][:, lead]
    if next_expected is not None and steps_done < episode_rows:
        ne = torch.as_tensor(next_expected, dtype=kb.dtype, device=kb.device)[:, None]
        kb[P.CLK_EXPECTED_ALL] += ne
        if next_outage is not None:
            kb[P.CLK_EXPECTED_OUTAGE] += ne * torch.as_tensor(next_outage, dtype=kb.dtype, device=kb.device)[:, None]
    building = {
        'electricity_consumption_total': _safe_div_t(kb[P.CLK_C_POS], kb[P.CLK_B_POS]),
        'zero_net_energy': _safe_div_t(kb[P.CLK_C_NET], kb[P.CLK_B_NET]),
        'carbon_emissions_total': _safe_div_t(kb[P.CLK_C_EMISSION], kb[P.CLK_B_EMISSION]),
        'cost_total': _safe_div_t(kb[P.CLK_C_COST], kb[P.CLK_B_COST]),
        'power_outage_normalized_unserved_energy_total': kb[P.CLK_UNSERVED_OUTAGE] / kb[P.CLK_EXPECTED_OUTAGE],
        'annual_normalized_unserved_energy_total': kb[P.CLK_UNSERVED_ALL] / kb[P.CLK_EXPECTED_ALL],
    }
    n = P.CLKE_PER_COND
    conds = []
    for c, extra_zero in ((0, False), (1, steps_done < episode_rows)):
        k = [kpi_env[c * n + j].double().clone() for j in range(n)]
        count = steps_done
        if extra_zero:
            _series_update(k, steps_done, torch.zeros_like(k[0]))
            count += 1
        # close the open groups with their actual sample count (pandas groupby mean / max of a partial group)
        out = {'ramp': k[P.CLKE_RAMP], 'all_max': k[P.CLKE_ALL_MAX]}
        for window, s_, m_, lf_, n_, key in ((24, P.CLKE_DAY_SUM, P.CLKE_DAY_MAX, P.CLKE_DAY_LF_SUM, P.CLKE_DAY_N, 'day'),
                                             (730, P.CLKE_MON_SUM, P.CLKE_MON_MAX, P.CLKE_MON_LF_SUM, P.CLKE_MON_N, 'mon')):
            open_cnt = count - window * ((count - 1) // window)
            groups = k[n_] + 1
            out[key + '_lf'] = (k[lf_] + 1 - (k[s_] / open_cnt) / k[m_]) / groups
            if key == 'day':
                out['day_peak'] = (k[P.CLKE_DAY_PEAK_SUM] + k[m_]) / groups
        conds.append(out)
    c, b = conds
    district = {
        'ramping_average': _safe_div_t(c['ramp'], b['ramp']),
        'daily_one_minus_load_factor_average': _safe_div_t(c['day_lf'], b['day_lf']),
        'monthly_one_minus_load_factor_average': _safe_div_t(c['mon_lf'], b['mon_lf']),
        'daily_peak_average': _safe_div_t(c['day_peak'], b['day_peak']),
        'all_time_peak_average': _safe_div_t(c['all_max'], b['all_max']),
    }
    for name, v in building.items():
        district[name] = torch.nanmean(v, dim=0)
    return building, district


COMFORT_KPIS = ('discomfort_proportion', 'discomfort_cold_proportion', 'discomfort_hot_proportion',
                'discomfort_cold_delta_minimum', 'discomfort_cold_delta_maximum', 'discomfort_cold_delta_average',
                'discomfort_hot_delta_minimum', 'discomfort_hot_delta_maximum', 'discomfort_hot_delta_average',
                'one_minus_thermal_resilience_proportion')


def finalize_comfort(kpi_comfort, spec, tables, steps_done: int, band: float = 2.0):
    """Discomfort KPIs of `CityLearnEnv.evaluate` (citylearn.py:1199-1215 -> cost_function.py:224-353) for every env from the
    device accumulators of the LSTM stage (`cl_lstm_step_f32`, `kpi_comfort [CL_NKC, n_bldg, n_env]`).  Like the reference's
    series, the statistics cover the `steps_done` simulated steps plus -- unless the data ran out -- the untouched data-file
    row of the next step; occupancy counts are env-independent.  Returns name -> tensor ``[n_bldg, n_env]``."""
    import torch
    P = abi
    k = kpi_comfort.double().clone()
    B = k.shape[1]
    n = min(steps_done + 1, tables.n_steps)
    occupied = torch.zeros((B, 1), dtype=k.dtype, device=k.device)
    occupied_outage = torch.zeros_like(occupied)
    for i, b in enumerate(spec.buildings):
        w = slice(tables.start, tables.start + n)
        occ = np.asarray(b.series['occupant_count'][w], dtype=np.float64) > 0.0
        out = np.asarray(tables.outage[:n, i]) != 0.0
        occupied[i] = occ.sum()
        occupied_outage[i] = (occ & out).sum()
        if n > steps_done:                                  # the extra row: data-file temperature, same for every env
            r = tables.start + steps_done
            temp = float(b.series['indoor_dry_bulb_temperature'][r])
            cd = temp - float(b.series['indoor_dry_bulb_temperature_cooling_set_point'][r]) if occ[-1] else 0.0
            hd = temp - float(b.series['indoor_dry_bulb_temperature_heating_set_point'][r]) if occ[-1] else 0.0
            hot, cold = cd > band, hd < -band
            cmag, hmag = abs(min(hd, 0.0)), abs(max(cd, 0.0))
            k[P.CLKC_UNMET, i] += float(hot or cold); k[P.CLKC_COLD, i] += float(cold); k[P.CLKC_HOT, i] += float(hot)
            k[P.CLKC_COLD_MIN, i] = torch.clamp(k[P.CLKC_COLD_MIN, i], max=cmag); k[P.CLKC_COLD_MAX, i] = torch.clamp(k[P.CLKC_COLD_MAX, i], min=cmag)
            k[P.CLKC_HOT_MIN, i] = torch.clamp(k[P.CLKC_HOT_MIN, i], max=hmag); k[P.CLKC_HOT_MAX, i] = torch.clamp(k[P.CLKC_HOT_MAX, i], min=hmag)
            k[P.CLKC_COLD_SUM, i] += cmag; k[P.CLKC_HOT_SUM, i] += hmag
            k[P.CLKC_UNMET_OUTAGE, i] += float((hot or cold) and out[-1])
    return {
        'discomfort_proportion': k[P.CLKC_UNMET] / occupied, 'discomfort_cold_proportion': k[P.CLKC_COLD] / occupied,
        'discomfort_hot_proportion': k[P.CLKC_HOT] / occupied,
        'discomfort_cold_delta_minimum': k[P.CLKC_COLD_MIN], 'discomfort_cold_delta_maximum': k[P.CLKC_COLD_MAX],
        'discomfort_cold_delta_average': k[P.CLKC_COLD_SUM] / n,
        'discomfort_hot_delta_minimum': k[P.CLKC_HOT_MIN], 'discomfort_hot_delta_maximum': k[P.CLKC_HOT_MAX],
        'discomfort_hot_delta_average': k[P.CLKC_HOT_SUM] / n,
        'one_minus_thermal_resilience_proportion': k[P.CLKC_UNMET_OUTAGE] / occupied_outage,
    }
