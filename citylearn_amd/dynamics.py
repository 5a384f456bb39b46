"""LSTM indoor-temperature stage of `LSTMDynamicsBuilding` on the GPU (adjacent to the energy step).

Host side: packs the per-building LSTM weights (``Building_k.pth``, reference `LSTMDynamics`, citylearn/dynamics.py:50-127)
and the env-independent part of the layer-0 gates per (t, building) -- eleven of the thirteen model inputs (weather,
calendar sin/cos, set point, occupancy; building.py:3057-3078, 1191-1201) do not depend on the env, so
``W_ih0[:, exo] @ x_exo(t) + b_ih0 + b_hh0`` is a table.  Device side: `cl_lstm_step_f32` (csrc/cl_lstm.h).
"""
from __future__ import annotations

import ctypes
from typing import List, Optional

import numpy as np
import torch

from . import _lib, abi
from .schema import DistrictSpec, EpisodeTables

_PERIODIC = {'month': 12, 'hour': 24, 'day_type': 7}     # building.py:1493-1498 (max of the ranges)

# offsets inside `lstm_w` (csrc/cl_lstm.h)
WC, WT, WHH0, WIH1, WHH1, B1, WLIN, BLIN, TMIN, TMAX, CMIN, CMAX, ACTIVE = 0, 64, 128, 1152, 2176, 3200, 3264, 3280, 3281, 3282, 3283, 3284, 3285
DEM_HEAT = 3290          # csrc/cl_lstm.h CLW_DEM_HEAT
DEM2, C2MIN, C2MAX = 3291, 3292, 3293      # CLW_DEM2: the model also takes heating_demand (second demand input) + its normalisation
W2 = 3296                                  # CLW_W2: W_ih0[:, second demand input] (matrix-core kernel)
PRE_TNORM, PRE_TRAW, PRE_HVAC, PRE_CSP, PRE_HSP, PRE_BAND, PRE_OCC, PRE_OUTAGE = 64, 65, 66, 67, 68, 69, 70, 71
RW_BAND, RW_LOEXP, RW_HIEXP, KPI_BAND = 3286, 3287, 3288, 3289


def _exo_feature(b, name: str, w: slice) -> np.ndarray:
    """Env-independent model input `name` over the episode window, as `Building.observations(periodic_normalization=True)`
    reports it (building.py:1191-1201; preprocessing.py:68-72)."""
    for base, x_max in _PERIODIC.items():
        if name in (f'{base}_sin', f'{base}_cos'):
            x = 2 * np.pi * b.series[base][w] / x_max
            return np.sin(x) if name.endswith('_sin') else np.cos(x)
    return np.asarray(b.series[name][w], dtype=np.float64)


# -log2(e) for the sigmoid gates i, f, o and -2 log2(e) for the tanh gate g, rows [i; f; g; o] x 16 (cl_lstm.h: lstm_act)
GATE_SCALE = -np.log2(np.e) * np.repeat([1.0, 1.0, 2.0, 1.0], 16)


def pack_lstm(spec: DistrictSpec, tables: EpisodeTables, band=None, lower_exponent: float = 2.0, higher_exponent: float = 2.0,
              kpi_band: float = 2.0):
    """Returns ``(lstm_w [B, CL_LSTM_NW] f32, dyn_pre [T, B, CL_LSTM_NPRE] f32)``; `band` / exponents are the
    ComfortReward parameters of the fused reward epilogue."""
    B, T = len(spec.buildings), tables.n_steps
    w = slice(tables.start, tables.end + 1)
    lstm_w = np.zeros((B, abi.CL_LSTM_NW), dtype=np.float32)
    dyn_pre = np.zeros((T, B, abi.CL_LSTM_NPRE), dtype=np.float32)
    for i, b in enumerate(spec.buildings):
        dyn_pre[:, i, PRE_TRAW] = b.series['indoor_dry_bulb_temperature'][w]
        dyn_pre[:, i, PRE_HVAC] = b.series['hvac_mode'][w]
        dyn_pre[:, i, PRE_CSP] = b.series['indoor_dry_bulb_temperature_cooling_set_point'][w]
        dyn_pre[:, i, PRE_HSP] = b.series['indoor_dry_bulb_temperature_heating_set_point'][w]
        dyn_pre[:, i, PRE_BAND] = b.series['comfort_band'][w]
        dyn_pre[:, i, PRE_OCC] = b.series['occupant_count'][w]
        dyn_pre[:, i, PRE_OUTAGE] = tables.outage[:, i]
        lstm_w[i, RW_BAND] = np.nan if band is None else band
        lstm_w[i, RW_LOEXP], lstm_w[i, RW_HIEXP] = lower_exponent, higher_exponent
        lstm_w[i, KPI_BAND] = kpi_band
        d = b.dynamics
        if d is None:
            continue
        H = d.hidden_size
        if d.lookback != 12 or d.num_layers not in (1, 2) or not 1 <= H <= abi.CL_LSTM_GEN_HMAX:
            raise NotImplementedError(f'LSTM dynamics need lookback 12, 1-2 layers and hidden size <= {abi.CL_LSTM_GEN_HMAX} '
                                      f'(got lookback {d.lookback}, {d.num_layers} layers, hidden size {H})')
        sd = torch.load(d.filepath, map_location='cpu')
        sd = {k: v.double().numpy() for k, v in sd.get('model_state_dict', sd).items()}
        names = list(d.input_observation_names)
        lo, hi = np.array(d.input_normalization_minimum, dtype=np.float64), np.array(d.input_normalization_maximum, dtype=np.float64)
        (ic, i2), it = _demand_inputs(names), names.index('indoor_dry_bulb_temperature')
        lstm_w[i, DEM_HEAT] = 1.0 if names[ic] == 'heating_demand' else 0.0
        if i2 is not None:                              # both demands among the inputs: delivered heating is the second env-dependent input
            lstm_w[i, DEM2], lstm_w[i, C2MIN], lstm_w[i, C2MAX] = 1.0, lo[i2], hi[i2]
        if _needs_generic_kernel(d):
            # another shape: the generic kernel's tables hold the weights (pack_lstm_generic); this row carries the
            # normalisation constants, the output bias and the kernel selector
            lstm_w[i, BLIN] = sd['l_linear.bias'].reshape(-1)[0]
            lstm_w[i, TMIN], lstm_w[i, TMAX] = lo[it], hi[it]
            lstm_w[i, CMIN], lstm_w[i, CMAX] = lo[ic], hi[ic]
            lstm_w[i, ACTIVE] = 2.0 if d.num_layers == 1 else 3.0
            dyn_pre[:, i, PRE_TNORM] = (np.asarray(b.series['indoor_dry_bulb_temperature'][w], dtype=np.float64) - lo[it]) / (hi[it] - lo[it])
            continue

        # A hidden size below 16 is embedded exactly: the padded units have zero weights and biases, so their cell and
        # hidden state stay 0 (c = 0.5 c + 0.5 tanh(0) = 0, h = 0.5 tanh(0) = 0) and nothing reads them.
        # The gate rows are pre-multiplied (in float64, one rounding to fp32) by -log2(e) (i, f, o) / -2 log2(e) (g): the kernel's
        # sigmoid / tanh are 1 / (1 + 2^z) and 2 / (1 + 2^z) - 1 of the accumulated z (one multiply per activation less).
        def gates(m):                                   # torch rows [i; f; g; o] x H  ->  scaled [i; f; g; o] x 16
            m = np.asarray(m, dtype=np.float64)
            out = np.zeros((4, 16) + m.shape[1:])
            out[:, :H] = m.reshape((4, H) + m.shape[1:])
            return out.reshape((64,) + m.shape[1:]) * GATE_SCALE.reshape((64,) + (1,) * (m.ndim - 1))

        def cols(m):                                    # [rows, H] -> [rows, 16]
            out = np.zeros((m.shape[0], 16))
            out[:, :H] = m
            return out

        wih0, whh0 = gates(sd['l_lstm.weight_ih_l0']), cols(gates(sd['l_lstm.weight_hh_l0']))
        b0 = gates(sd['l_lstm.bias_ih_l0'] + sd['l_lstm.bias_hh_l0'])
        lstm_w[i, WC:WC + 64] = wih0[:, ic]
        lstm_w[i, WT:WT + 64] = wih0[:, it]
        if i2 is not None:
            lstm_w[i, W2:W2 + 64] = wih0[:, i2]          # rides in the idle k-slot of the pre-gate product (csrc/cl_lstm.h)
        lstm_w[i, WHH0:WHH0 + 1024] = whh0.reshape(-1)
        lstm_w[i, WIH1:WIH1 + 1024] = cols(gates(sd['l_lstm.weight_ih_l1'])).reshape(-1)
        lstm_w[i, WHH1:WHH1 + 1024] = cols(gates(sd['l_lstm.weight_hh_l1'])).reshape(-1)
        lstm_w[i, B1:B1 + 64] = gates(sd['l_lstm.bias_ih_l1'] + sd['l_lstm.bias_hh_l1'])
        lstm_w[i, WLIN:WLIN + 16] = cols(sd['l_linear.weight'].reshape(1, -1)).reshape(-1)
        lstm_w[i, BLIN] = sd['l_linear.bias'].reshape(-1)[0]
        lstm_w[i, TMIN], lstm_w[i, TMAX] = lo[it], hi[it]
        lstm_w[i, CMIN], lstm_w[i, CMAX] = lo[ic], hi[ic]
        lstm_w[i, ACTIVE] = 1.0
        pre = np.tile(b0[None, :], (T, 1))
        for k, name in enumerate(names):
            if k in (ic, it, i2):
                continue
            x = (_exo_feature(b, name, w) - lo[k]) / (hi[k] - lo[k])
            pre += x[:, None] * wih0[None, :, k]
        dyn_pre[:, i, :64] = pre
        dyn_pre[:, i, PRE_TNORM] = (np.asarray(b.series['indoor_dry_bulb_temperature'][w], dtype=np.float64) - lo[it]) / (hi[it] - lo[it])
    return lstm_w, dyn_pre


def cell_update_bounds(spec: DistrictSpec, tables: EpisodeTables, lstm_w: np.ndarray, dyn_pre: np.ndarray):
    """Rigorous upper bounds of the (pre-scaled) gate values z of every matrix-core LSTM of the district: ``(worst z_i + z_f + z_g over the
    hidden units of both layers, worst z_o)``.  The common-denominator cell update of csrc/cl_lstm.h (7 instead of 10 transcendentals per
    unit and cell) multiplies (1 + 2^z_i)(1 + 2^z_g)(1 + 2^z_f) and (1 + 2^z_o)(1 + 2^z_c), z_c <= 64: finite in fp32 while
    z_i + z_f + z_g < 126 and z_o < 62.  What the gates can see is bounded by construction: the exogenous part is the table `dyn_pre`, a
    hidden state lies in (-1, 1), the temperature input is a data-file value (warm-up) or a model output b + sum |W_lin| at most, and the
    demand input is what the building's device and tank can deliver."""
    worst_sum, worst_o = -np.inf, -np.inf
    w = slice(tables.start, tables.end + 1)
    for i, b in enumerate(spec.buildings):
        W = lstm_w[i].astype(np.float64)
        if W[ACTIVE] != 1.0:
            continue
        heat = W[DEM_HEAT] != 0.0
        dev, tank, series = (b.heating_device, b.heating_storage, 'heating_demand') if heat else (b.cooling_device, b.cooling_storage, 'cooling_demand')
        t_out = np.asarray(b.series['outdoor_dry_bulb_temperature'][w], dtype=np.float64)
        cop = np.max(dev.cop(t_out, heating=heat)) if getattr(dev, 'is_heat_pump', True) else float(dev.efficiency)
        dem_hi = max(float(np.max(b.series[series][w])), float(dev.nominal_power) * float(cop)) + float(tank.capacity)
        span_c = W[CMAX] - W[CMIN]
        xc = np.array([(0.0 - W[CMIN]) / span_c, (dem_hi - W[CMIN]) / span_c])
        t_model = W[BLIN] + np.array([-1.0, 1.0]) * np.abs(W[WLIN:WLIN + 16]).sum()
        t_file = dyn_pre[:, i, PRE_TNORM].astype(np.float64)
        xt = np.array([min(t_model[0], t_file.min()), max(t_model[1], t_file.max())])
        pre = dyn_pre[:, i, :64].astype(np.float64).max(axis=0)
        z0 = pre + np.max(np.outer(W[WC:WC + 64], xc), axis=1) + np.max(np.outer(W[WT:WT + 64], xt), axis=1) \
            + np.abs(W[WHH0:WHH0 + 1024].reshape(64, 16)).sum(axis=1)
        if W[DEM2] != 0.0:                               # a second demand input (delivered heating)
            hd, hs = b.heating_device, b.heating_storage
            cop_h = np.max(hd.cop(t_out, heating=True)) if getattr(hd, 'is_heat_pump', False) else float(hd.efficiency)
            h_hi = max(float(np.max(b.series['heating_demand'][w])), float(hd.nominal_power) * float(cop_h)) + float(hs.capacity)
            span2 = W[C2MAX] - W[C2MIN]
            z0 = z0 + np.max(np.outer(W[W2:W2 + 64], np.array([(0.0 - W[C2MIN]) / span2, (h_hi - W[C2MIN]) / span2])), axis=1)
        z1 = W[B1:B1 + 64] + np.abs(W[WIH1:WIH1 + 1024].reshape(64, 16)).sum(axis=1) + np.abs(W[WHH1:WHH1 + 1024].reshape(64, 16)).sum(axis=1)
        for z in (z0, z1):
            zi, zf, zg, zo = z.reshape(4, 16)
            worst_sum = max(worst_sum, float(np.max(np.maximum(zi, 0.0) + np.maximum(zf, 0.0) + np.maximum(zg, 0.0))))
            worst_o = max(worst_o, float(zo.max()))
    return worst_sum, worst_o


def _demand_inputs(names):
    """``(index of the model's demand input, index of a second one or None)``: `cooling_demand`, or `heating_demand` for a heating-driven
    model ("LSTM model only uses either cooling/heating demand not both as input variable", building.py:3013-3017) -- or both: the
    reference builds the model input generically from `input_observation_names` (building.py:3039-3078), so a model trained on both
    demands runs there; here it takes cooling as the first and heating as the second env-dependent input of the generic kernel."""
    have = [n for n in ('cooling_demand', 'heating_demand') if n in names]
    if not have:
        raise NotImplementedError('LSTM dynamics need cooling_demand and / or heating_demand among their inputs')
    return names.index(have[0]), (names.index(have[1]) if len(have) == 2 else None)


FORCE_GENERIC_KERNEL = False          # tests: route every model to cl_lstm_generic_kernel (cross-checks the two kernels on the same fixtures)


def _needs_generic_kernel(d) -> bool:
    """The matrix-core kernel covers two layers of <= 16 units (one or both demand inputs); everything else runs on cl_lstm_generic_kernel."""
    return FORCE_GENERIC_KERNEL or not (d.num_layers == 2 and d.hidden_size <= 16)


def pack_lstm_generic(spec: DistrictSpec, tables: EpisodeTables):
    """Tables of `cl_lstm_generic_step_f32` for the buildings `pack_lstm` marked ACTIVE = 2 / 3 (LSTM shapes other than two
    layers of <= 16 units): ``(gen_w [B, GW] f32, gen_pre [T, B, H, 4] f32, H)`` with H the largest hidden size among them, or
    ``None`` when there is no such building.  Layouts: csrc/cl_lstm.h (gate order i, f, g, o)."""
    todo = [(i, b) for i, b in enumerate(spec.buildings) if b.dynamics is not None and _needs_generic_kernel(b.dynamics)]
    if not todo:
        return None
    B, T = len(spec.buildings), tables.n_steps
    w = slice(tables.start, tables.end + 1)
    H = max(b.dynamics.hidden_size for _, b in todo)
    gw = H * 12 + 3 * H * H * 4 + H * 4 + H
    gen_w = np.zeros((B, gw), dtype=np.float32)
    gen_pre = np.zeros((T, B, H, 4), dtype=np.float32)
    o_wx, o_hh0 = 0, H * 12
    o_ih1, o_hh1 = o_hh0 + H * H * 4, o_hh0 + 2 * H * H * 4
    o_b1 = o_hh0 + 3 * H * H * 4
    o_lin = o_b1 + H * 4
    for i, b in todo:
        d = b.dynamics
        h = d.hidden_size
        sd = torch.load(d.filepath, map_location='cpu')
        sd = {k: v.double().numpy() for k, v in sd.get('model_state_dict', sd).items()}
        names = list(d.input_observation_names)
        lo, hi = np.array(d.input_normalization_minimum, dtype=np.float64), np.array(d.input_normalization_maximum, dtype=np.float64)
        (ic, i2), it = _demand_inputs(names), names.index('indoor_dry_bulb_temperature')

        def ug(m):                                      # torch rows [i; f; g; o] x h (x cols)  ->  [unit, gate, (cols)], padded to H units
            m = np.asarray(m, dtype=np.float64).reshape((4, h) + np.shape(m)[1:])
            out = np.zeros((H, 4) + m.shape[2:])
            out[:h] = np.moveaxis(m, 0, 1)
            return out

        def square(m):                                  # [4h, h] -> [unit u][input k][gate]
            t = ug(m)                                   # [H, 4, h]
            out = np.zeros((H, H, 4))
            out[:, :h, :] = np.moveaxis(t, 1, 2)
            return out

        wih0 = ug(sd['l_lstm.weight_ih_l0'])            # [H, 4, n_in]
        w2 = wih0[:, :, i2] if i2 is not None else np.zeros_like(wih0[:, :, ic])
        gen_w[i, o_wx:o_wx + H * 12] = np.concatenate([wih0[:, :, ic], wih0[:, :, it], w2], axis=1).reshape(-1)
        gen_w[i, o_hh0:o_hh0 + H * H * 4] = square(sd['l_lstm.weight_hh_l0']).reshape(-1)
        if d.num_layers == 2:
            gen_w[i, o_ih1:o_ih1 + H * H * 4] = square(sd['l_lstm.weight_ih_l1']).reshape(-1)
            gen_w[i, o_hh1:o_hh1 + H * H * 4] = square(sd['l_lstm.weight_hh_l1']).reshape(-1)
            gen_w[i, o_b1:o_b1 + H * 4] = ug(sd['l_lstm.bias_ih_l1'] + sd['l_lstm.bias_hh_l1']).reshape(-1)
        gen_w[i, o_lin:o_lin + h] = sd['l_linear.weight'].reshape(-1)
        pre = np.tile(ug(sd['l_lstm.bias_ih_l0'] + sd['l_lstm.bias_hh_l0'])[None], (T, 1, 1))        # [T, H, 4]
        for k, name in enumerate(names):
            if k in (ic, it, i2):
                continue
            x = (_exo_feature(b, name, w) - lo[k]) / (hi[k] - lo[k])
            pre += x[:, None, None] * wih0[None, :, :, k]
        gen_pre[:, i] = pre
    return gen_w, gen_pre, H


def _bf16_split3(x: np.ndarray) -> np.ndarray:
    """x (float64 / float32) -> [3, ...] uint16: three round-to-nearest-even bf16 terms with x ~= t0 + t1 + t2."""
    r = np.asarray(x, dtype=np.float32).copy()
    out = []
    for _ in range(3):
        bits = r.view(np.uint32).astype(np.uint64)
        q = ((bits + 0x7FFF + ((bits >> 16) & 1)) >> 16).astype(np.uint32)          # RNE to the upper 16 bits
        out.append(q.astype(np.uint16))
        r = (r - (q << 16).astype(np.uint32).view(np.float32)).astype(np.float32)    # exact residual
    return np.stack(out)


def _f16_split2(x: np.ndarray) -> np.ndarray:
    """x -> [2, ...] uint16: two round-to-nearest-even f16 terms with x ~= t0 + t1 (|x - t0 - t1| <= 2^-22 |x|, or 2^-25 absolute
    where t1 is subnormal)."""
    r = np.asarray(x, dtype=np.float32)
    if np.abs(r).max(initial=0.0) >= 65504.0:
        raise ValueError('LSTM weight outside the f16 range: use the bf16 split')
    t0 = r.astype(np.float16)
    t1 = (r - t0.astype(np.float32)).astype(np.float16)          # the residual is exact in fp32
    return np.stack([t0.view(np.uint16), t1.view(np.uint16)])


def pack_lstm_split(lstm_w: np.ndarray, fmt: str = 'bf16') -> np.ndarray:
    """The three recurrent matrices of `lstm_w` ([B, CL_LSTM_NW], padded 64 x 16 layout) as split 16-bit MFMA A-operand
    fragments: ``[B, 18, 64, 8]`` uint16 (the CL_LSTM_NWB stride of the C-ABI), fragment ``(2 * matrix{hh0, ih1, hh1} + row_block) * T
    + term`` with T = 3 bf16 terms (`fmt` 'bf16') or 2 f16 terms ('f16': the first 12 fragments, CLD_LSTM_F16), element
    ``[lane][j] = W[32 row_block + (lane & 31)][unit (j & 3) + 8 (j >> 2) + 4 (lane >> 5)]`` (csrc/cl_lstm.h)."""
    B = lstm_w.shape[0]
    out = np.zeros((B, 18, 64, 8), dtype=np.uint16)
    lane = np.arange(64)
    j = np.arange(8)
    unit = (j[None, :] & 3) + 8 * (j[None, :] >> 2) + 4 * (lane[:, None] >> 5)      # [64, 8]
    T, split_fn = {'bf16': (3, _bf16_split3), 'f16': (2, _f16_split2)}[fmt]
    for m, base in enumerate((WHH0, WIH1, WHH1)):
        Wm = lstm_w[:, base:base + 1024].reshape(B, 64, 16)
        for rb in range(2):
            rows = 32 * rb + (lane & 31)                                             # [64]
            frag = Wm[:, rows[:, None], unit]                                        # [B, 64, 8]
            split = split_fn(frag)                                                   # [T, B, 64, 8]
            for k in range(T):
                out[:, (m * 2 + rb) * T + k] = split[k]
    return out


class LSTMStage:
    """Device state + driver of the LSTM stage for one env shard (pairs with a `StepEngine` built with detail=True)."""

    def __init__(self, spec: DistrictSpec, tables: EpisodeTables, engine, band=None, lower_exponent: float = 2.0,
                 higher_exponent: float = 2.0, kpi: bool = False, kpi_band: float = 2.0, split: Optional[str] = 'f16',
                 cell_update: str = 'auto'):
        """`kpi`: accumulate the discomfort KPIs on the device (`kpi_comfort`, finalised by `kpi.finalize_comfort`) with the
        scalar comfort band `kpi_band` (`CityLearnEnv.evaluate`'s ``comfort_band``, default 2.0 C -- data.py:399).
        `split`: operand format of the recurrent products on the matrix cores -- 'f16' (two terms per operand, three partial
        products: the default), 'bf16' (three terms, six partial products: twice the matrix-pipe time for dropped terms of
        2^-24 instead of 3 * 2^-22 relative) or None (exact f32 MFMA).
        `cell_update`: 'plain' (sigmoid / tanh per gate: 10 transcendentals per hidden unit and cell), 'common_denominator' (7, csrc/cl_lstm.h:
        same values to a few fp32 roundings; needs bounded gate values) or 'auto' -- the second form where `cell_update_bounds` proves the
        bound for every model of the district (all 2023 models; not baeda_3dem's), else the first.  An explicit
        ``tuning={'lstm_variant': ...}`` of the engine wins."""
        self.lib = _lib.load()
        self.engine = engine
        self._args = None
        lstm_w, dyn_pre = pack_lstm(spec, tables, band, lower_exponent, higher_exponent, kpi_band)
        self.kpi_band = kpi_band
        dev = engine.device
        self.any_active = bool(lstm_w[:, ACTIVE].any())
        self.lstm_w = torch.from_numpy(lstm_w).to(dev)
        # split 16-bit fragments of the recurrent matrices (matrix-core path); int16 storage of the raw bits
        self.lstm_wb = torch.from_numpy(pack_lstm_split(lstm_w, split).view(np.int16)).to(dev) if split else None
        self.dims = _lib.Dims.from_buffer_copy(engine.dims)          # the engine's dims (never mutated) + the operand-format flag
        if split == 'f16':
            self.dims.flags |= abi.CLD_LSTM_F16
        if bool(((lstm_w[:, ACTIVE] == 1.0) & (lstm_w[:, DEM2] != 0.0)).any()):
            self.dims.flags |= abi.CLD_LSTM_TWO_DEMANDS          # a 2 x 16-unit model that takes both demands: the instantiation with the third input ring
        # the stage's own copy of the launch overrides: the cell-update form is chosen here
        self.tuning = _lib.Tuning.from_buffer_copy(engine.tuning)
        self.dims.tuning = ctypes.pointer(self.tuning)
        self.cell_bounds = cell_update_bounds(spec, tables, lstm_w, dyn_pre) if self.any_active else (-np.inf, -np.inf)
        admitted = self.cell_bounds[0] < 126.0 and self.cell_bounds[1] < 62.0
        if cell_update not in ('auto', 'plain', 'common_denominator'):
            raise ValueError("cell_update must be 'auto', 'plain' or 'common_denominator'")
        if cell_update == 'common_denominator' and not admitted:
            raise ValueError(f'the common-denominator cell update needs z_i + z_f + z_g < 126 and z_o < 62; this district reaches {self.cell_bounds}')
        self.cell_update = 'plain'
        if self.tuning.lstm_variant == 0 and split is not None and cell_update != 'plain' and admitted:
            self.tuning.lstm_variant = 32
            self.cell_update = 'common_denominator'
        self.dyn_pre = torch.from_numpy(dyn_pre).to(dev)
        B, E = engine.n_bldg, engine.n_env
        self.hist = torch.zeros((abi.CL_LSTM_NHIST, B, E), dtype=torch.float32, device=dev)
        self.hidden = torch.zeros((B, E, abi.CL_LSTM_NHIDDEN), dtype=torch.float32, device=dev)
        self.indoor_temp = torch.zeros((B, E), dtype=torch.float32, device=dev)
        self.comfort = torch.zeros((B, E), dtype=torch.float32, device=dev)
        self.kpi_comfort = torch.zeros((abi.CL_NKC, B, E), dtype=torch.float32, device=dev) if kpi else None
        # buildings with another LSTM shape (ACTIVE = 2 / 3): tables and carried state of the generic kernel
        self.generic = None
        packed = pack_lstm_generic(spec, tables)
        if packed is not None:
            gen_w, gen_pre, gen_h = packed
            self.generic = dict(w=torch.from_numpy(gen_w).to(dev), pre=torch.from_numpy(gen_pre).to(dev), h=int(gen_h),
                                layers=2 if bool((lstm_w[:, ACTIVE] == 3.0).any()) else 1,
                                hidden=torch.zeros((B, 4, gen_h, E), dtype=torch.float32, device=dev))
            self.lib.cl_lstm_generic_step_f32.argtypes = [ctypes.POINTER(_lib.Dims)] + [ctypes.c_void_p] * 3 + [ctypes.c_int64] + [
                ctypes.c_void_p] * 2 + [ctypes.c_int32, ctypes.c_int32] + [ctypes.c_void_p] * 6 + [ctypes.c_int32, ctypes.c_void_p]
        self.lib.cl_lstm_step_f32.argtypes = [ctypes.POINTER(_lib.Dims)] + [ctypes.c_void_p] * 10 + [ctypes.c_int32, ctypes.c_void_p]
        self.lib.cl_lstm_reset_f32.argtypes = [ctypes.POINTER(_lib.Dims)] + [ctypes.c_void_p] * 4
        self.reset()

    def _stream(self) -> int:
        return torch.cuda.current_stream(self.engine.device).cuda_stream

    def reset(self):
        with torch.cuda.device(self.engine.device):
            _lib.check(self.lib.cl_lstm_reset_f32(ctypes.byref(self.engine.dims), self.hist.data_ptr(), self.hidden.data_ptr(),
                                                  None if self.kpi_comfort is None else self.kpi_comfort.data_ptr(), self._stream()))
            if self.generic is not None:
                self.generic['hidden'].zero_()
            self.indoor_temp.zero_()            # handed out by the 'planes' observation of reset()
            self.comfort.zero_()

    def _carried(self):
        return [('hist', self.hist), ('hidden', self.hidden), ('indoor_temp', self.indoor_temp), ('comfort', self.comfort),
                ('kpi_comfort', self.kpi_comfort), ('generic_hidden', None if self.generic is None else self.generic['hidden'])]

    def state_dict(self) -> dict:
        """What the temperature stage carries between steps: the 12-row input history ring, the LSTM hidden / cell state, the last indoor
        temperature and comfort reward, the comfort KPI accumulators (`StepEngine.state_dict` is the energy side)."""
        return {k: None if v is None else v.detach().clone() for k, v in self._carried()}

    def load_state_dict(self, sd) -> None:
        for k, dst in self._carried():
            src = sd.get(k)
            if (src is None) != (dst is None) or (src is not None and tuple(src.shape) != tuple(dst.shape)):
                raise ValueError(f'checkpoint tensor {k!r} does not fit this LSTM stage')
            if dst is not None:
                dst.copy_(src.to(dst.device))

    def step(self, t: int, cool_dem: Optional[torch.Tensor] = None, heat_dem: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Call right after ``engine.step(actions, t)``.  Returns the indoor temperature ``[n_bldg, n_env]`` of step t.
        ``cool_dem`` / ``heat_dem``: delivered cooling / heating planes to use instead of the engine's own (tests)."""
        e = self.engine
        if self._args is None:          # per-step arguments that never change, converted once
            self._args = ((ctypes.byref(self.dims), self.lstm_w.data_ptr(), None if self.lstm_wb is None else self.lstm_wb.data_ptr(),
                           self.dyn_pre.data_ptr()),
                          (self.hist.data_ptr(), self.hidden.data_ptr(), self.indoor_temp.data_ptr(), self.comfort.data_ptr(),
                           None if self.kpi_comfort is None else self.kpi_comfort.data_ptr()),
                          e.out_bldg[abi.CLO_COOL_DEM].data_ptr(), e.out_bldg[abi.CLO_HEAT_DEM].data_ptr())
        head, tail, own_cd, own_hd = self._args
        # delivered heating: the engine's own plane (written with the detail planes), or the caller's; with a caller-fed cooling
        # plane and no heating plane the heating side is zero (cooling-only tests)
        hd = heat_dem.data_ptr() if heat_dem is not None else (own_hd if cool_dem is None else None)
        tail = (hd,) + tail
        self.tuning.kernel_name = e.tuning.kernel_name               # (diagnostics: StepEngine.trace_kernels)
        with e._on_device():
            rc = self.lib.cl_lstm_step_f32(*head, own_cd if cool_dem is None else cool_dem.data_ptr(), *tail, int(t), e._stream())
            if not rc and self.generic is not None:
                gn = self.generic
                rc = self.lib.cl_lstm_generic_step_f32(head[0], head[1], head[3], gn['w'].data_ptr(), gn['w'].shape[1], gn['pre'].data_ptr(),
                                                       gn['hidden'].data_ptr(), gn['h'], gn['layers'], own_cd if cool_dem is None else cool_dem.data_ptr(),
                                                       hd, tail[1], tail[3], tail[4], tail[5], int(t), e._stream())
        if rc:
            _lib.check(rc)
        return self.indoor_temp
