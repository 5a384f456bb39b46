"""The packed LSTM tables (`dynamics.pack_lstm` + `pack_lstm_split`) run through a numpy restatement of cl_lstm_kernel's window loop
(no GPU): what the device reads -- pre-scaled gate rows, host pre-gates, split 16-bit weight fragments -- must reproduce the
reference's predicted temperatures.  Catches packing mistakes (gate scaling, fragment order, ring semantics) without a GPU box."""
import numpy as np
import pytest

from citylearn_amd import dynamics as dyn
from golden_util import golden


def _weights_from_fragments(lstm_wb, fmt):
    """[18, 64, 8] uint16 fragments of one building -> the three 64 x 16 matrices the matrix cores see (sum of the split terms)."""
    T = 3 if fmt == 'bf16' else 2
    lane, j = np.arange(64), np.arange(8)
    unit = (j[None, :] & 3) + 8 * (j[None, :] >> 2) + 4 * (lane[:, None] >> 5)
    mats = []
    for m in range(3):
        W = np.zeros((64, 16))
        for rb in range(2):
            frag = np.zeros((64, 8))
            for k in range(T):
                bits = lstm_wb[(m * 2 + rb) * T + k]
                frag += ((bits.astype(np.uint32) << 16).view(np.float32) if fmt == 'bf16' else bits.view(np.float16)).astype(np.float64)
            W[(32 * rb + (lane & 31))[:, None], unit] = frag
        mats.append(W)
    return mats


def _cell(z, c):
    """cl_lstm.h lstm_act: the gate rows arrive pre-multiplied by -log2 e (i, f, o) / -2 log2 e (g)."""
    i, f = 1.0 / (1.0 + np.exp2(z[0:16])), 1.0 / (1.0 + np.exp2(z[16:32]))
    g, o = 2.0 / (1.0 + np.exp2(z[32:48])) - 1.0, 1.0 / (1.0 + np.exp2(z[48:64]))
    c = f * c + i * g
    return c, o * np.tanh(c)


@pytest.mark.parametrize('name,fmt,steps', [('g2023_p2', 'f16', 120), ('g2023_p2', 'bf16', 60), ('g2023_heat', 'f16', 120), ('g2023_both', 'f16', 120)])
def test_packed_tables_reproduce_the_reference_temperatures(name, fmt, steps):
    g = golden(name)
    spec = g.spec()
    tab = spec.episode_tables(0)
    lstm_w, dyn_pre = dyn.pack_lstm(spec, tab)
    wb = dyn.pack_lstm_split(lstm_w, fmt)
    B = lstm_w.shape[0]
    cool = g.ref['cool_dem']
    heat = g.ref['heat_dem'] if 'heat_dem' in g.ref.files else np.zeros_like(cool)
    steps = min(steps, g.facts['steps'])
    worst = 0.0
    for b in range(B):
        w = lstm_w[b].astype(np.float64)
        assert w[dyn.ACTIVE] == 1.0
        whh0, wih1, whh1 = _weights_from_fragments(wb[b], fmt)
        # the fragments are the (scaled) fp32 matrices of lstm_w, split
        np.testing.assert_allclose(whh0, w[dyn.WHH0:dyn.WHH0 + 1024].reshape(64, 16), rtol=2.0 ** -21, atol=2.0 ** -24)
        wc, wt, b1 = w[dyn.WC:dyn.WC + 64], w[dyn.WT:dyn.WT + 64], w[dyn.B1:dyn.B1 + 64]
        w2 = w[dyn.W2:dyn.W2 + 64]                       # second demand input (g2023_both's Building_1; zeros elsewhere)
        assert bool(np.any(w2)) == (w[dyn.DEM2] != 0.0) == (name == 'g2023_both' and b == 0)
        wlin, blin = w[dyn.WLIN:dyn.WLIN + 16], w[dyn.BLIN]
        tmin, tmax, cmin, cmax = w[dyn.TMIN], w[dyn.TMAX], w[dyn.CMIN], w[dyn.CMAX]
        dem = heat[:, b] if w[dyn.DEM_HEAT] != 0.0 else cool[:, b]
        ring_c, ring_t, ring_2 = np.zeros(12), np.zeros(12), np.zeros(12)
        h0 = np.zeros(16); c0 = np.zeros(16); h1 = np.zeros(16); c1 = np.zeros(16)
        for t in range(steps):
            ring_c[t % 12] = (dem[t] - cmin) / (cmax - cmin)                    # building.py:3068-3078
            if w[dyn.DEM2] != 0.0:
                ring_2[t % 12] = (heat[t, b] - w[dyn.C2MIN]) / (w[dyn.C2MAX] - w[dyn.C2MIN])
            y = dyn_pre[t, b, dyn.PRE_TNORM]
            temp = dyn_pre[t, b, dyn.PRE_TRAW]
            if t >= 12:                                                          # lookback + 1 samples exist (building.py:2996-2999)
                for s in range(12):
                    time = t - 11 + s
                    z0 = dyn_pre[time, b, :64].astype(np.float64) + wc * ring_c[time % 12] + wt * ring_t[(time - 1) % 12] + w2 * ring_2[time % 12] + whh0 @ h0
                    c0, h0 = _cell(z0, c0)
                    z1 = b1 + wih1 @ h0 + whh1 @ h1
                    c1, h1 = _cell(z1, c1)
                y = blin + wlin @ h1
                temp = y * (tmax - tmin) + tmin                                 # building.py:3031-3037
            ring_t[t % 12] = y                                                  # building.py:3027-3028
            worst = max(worst, abs(temp - g.ref['indoor_temp'][t][b]))
    assert worst < 5e-5, worst          # deg C (measured 5e-6); the device test's bound is 2e-3 (fp32 kernel), this restatement runs in float64


def test_cell_update_bounds_admit_the_2023_models_and_refuse_baeda():
    """`dynamics.cell_update_bounds`: the rigorous bound on the gate values that lets a district use the common-denominator cell update
    (z_i + z_f + z_g < 126, z_o < 62 keep its products finite in fp32)."""
    from citylearn_amd.dynamics import cell_update_bounds, pack_lstm
    for name, ok in (('g2023_p2', True), ('s_2023_p3', True), ('g2023_heat', True), ('s_baeda', False)):
        spec = golden(name).spec()
        tab = spec.episode_tables(0)
        w, pre = pack_lstm(spec, tab)
        zs, zo = cell_update_bounds(spec, tab, w, pre)
        assert 20.0 < zs < 200.0 and 5.0 < zo < 62.0
        assert (zs < 126.0) == ok, (name, zs, zo)


@pytest.mark.parametrize('name,b', [('g2023_both', 0), ('s_baeda', 3), ('g2023_p2', 1)])
def test_generic_kernel_tables_reproduce_the_reference_temperatures(name, b, monkeypatch):
    """`dynamics.pack_lstm_generic` (cl_lstm_generic_kernel's tables: WX [H][12] = gates of the demand, the temperature and the second demand
    input, square matrices as [unit][input][gate], host pre-gates) through a numpy restatement of that kernel's window loop: g2023_both's
    Building_1 takes BOTH demands (delivered heating as a third env-dependent input with its own ring), baeda's Building_4 is LSTM(11 -> 50, one layer)."""
    monkeypatch.setattr(dyn, 'FORCE_GENERIC_KERNEL', True)        # (2 x 16-unit models normally run on the matrix-core kernel)
    g = golden(name)
    spec = g.spec()
    tab = spec.episode_tables(0)
    lstm_w, dyn_pre = dyn.pack_lstm(spec, tab)
    gen_w, gen_pre, H = dyn.pack_lstm_generic(spec, tab)
    w = lstm_w[b].astype(np.float64)
    layers = {2.0: 1, 3.0: 2}[w[dyn.ACTIVE]]
    G = gen_w[b].astype(np.float64)
    wx = G[:H * 12].reshape(H, 3, 4)
    o = H * 12
    whh0 = G[o:o + H * H * 4].reshape(H, H, 4); o += H * H * 4
    wih1 = G[o:o + H * H * 4].reshape(H, H, 4); o += H * H * 4
    whh1 = G[o:o + H * H * 4].reshape(H, H, 4); o += H * H * 4
    b1 = G[o:o + H * 4].reshape(H, 4); o += H * 4
    wlin = G[o:o + H]
    two = w[dyn.DEM2] != 0.0
    assert two == (name == 'g2023_both')
    cool = g.ref['cool_dem'][:, b]
    heat = g.ref['heat_dem'][:, b] if 'heat_dem' in g.ref.files else np.zeros_like(cool)
    dem = heat if w[dyn.DEM_HEAT] != 0.0 else cool
    sig = lambda x: 1.0 / (1.0 + np.exp(-x))

    def cell(z, c):                                     # z [H, 4] = gates i, f, g, o (unscaled: the generic tables keep torch's values)
        c = sig(z[:, 1]) * c + sig(z[:, 0]) * np.tanh(z[:, 2])
        return c, sig(z[:, 3]) * np.tanh(c)

    ring_c, ring_t, ring_2 = np.zeros(12), np.zeros(12), np.zeros(12)
    h0 = np.zeros(H); c0 = np.zeros(H); h1 = np.zeros(H); c1 = np.zeros(H)
    worst = 0.0
    for t in range(min(100, g.facts['steps'])):
        ring_c[t % 12] = (dem[t] - w[dyn.CMIN]) / (w[dyn.CMAX] - w[dyn.CMIN])
        if two:
            ring_2[t % 12] = (heat[t] - w[dyn.C2MIN]) / (w[dyn.C2MAX] - w[dyn.C2MIN])
        y, temp = dyn_pre[t, b, dyn.PRE_TNORM], dyn_pre[t, b, dyn.PRE_TRAW]
        if t >= 12:
            for s in range(12):
                time = t - 11 + s
                z0 = gen_pre[time, b].astype(np.float64) + wx[:, 0] * ring_c[time % 12] + wx[:, 1] * ring_t[(time - 1) % 12] + wx[:, 2] * ring_2[time % 12] \
                    + np.einsum('ukg,k->ug', whh0, h0)
                c0, h0 = cell(z0, c0)
                if layers == 2:
                    z1 = b1 + np.einsum('ukg,k->ug', wih1, h0) + np.einsum('ukg,k->ug', whh1, h1)
                    c1, h1 = cell(z1, c1)
            y = w[dyn.BLIN] + wlin @ (h1 if layers == 2 else h0)
            temp = y * (w[dyn.TMAX] - w[dyn.TMIN]) + w[dyn.TMIN]
        ring_t[t % 12] = y
        worst = max(worst, abs(temp - g.ref['indoor_temp'][t][b]))
    assert worst < 5e-5, worst
