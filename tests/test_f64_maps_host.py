"""CLD_F64_MAPS on the CPU: `cl::unit_step<FULL, F64>` of csrc/cl_unit.h compiled with g++ (tests/host_shim, a test harness -- the product
has no CPU path) and run FREE-RUNNING over whole reference fixtures.  `battery_charge_ref` follows the reference's Battery.charge
operation by operation in the reference's own mixed precision (float64 with the float32 operations numpy's promotion rules put in it,
float32 rounding where the float32 series round, efficiency / degraded capacity carried as hi + lo pairs), so the battery state the
unit feeds back to itself is BIT-IDENTICAL to the reference's for every step of every fixture, outage rows included, and everything
else stays inside 0.07 x (1e-4 + 1e-4 |ref|).  (The fp32 map drifts to ~2e-4 relative on the 2020 fixture: DESIGN.md section 3.)
The GPU kernels run the same header (tests/test_gpu_parity.py::test_free_running_whole_fixture_f64)."""
import ctypes
import subprocess
from pathlib import Path

import numpy as np
import pytest

from golden_util import golden
from citylearn_amd import abi

HERE = Path(__file__).resolve().parent / 'host_shim'


@pytest.fixture(scope='module')
def shim(tmp_path_factory):
    out = tmp_path_factory.mktemp('shim') / 'libcl_unit_host.so'
    subprocess.run(['g++', '-O2', '-shared', '-fPIC', '-DCL_HOST_SHIM', '-ffp-contract=off', str(HERE / 'cl_unit_host.cpp'), '-o', str(out)], check=True)
    lib = ctypes.CDLL(str(out))
    vp = ctypes.c_void_p
    lib.host_unit_step.argtypes = [vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp, vp, vp]
    return lib


def free_run(lib, name: str, f64):
    """Worst error / (1e-4 + 1e-4 |ref|) of soc, efficiency, degraded capacity, tank SoCs and net over the fixture, free-running.
    `f64`: False / 0 = fp32 battery map, True / 1 = CLD_F64_MAPS, 2 = CLD_F64_CHAIN (the degraded-capacity plane holds the capacity loss)."""
    f64 = int(f64)
    g = golden(name)
    spec = g.spec()
    tab = spec.episode_tables(0)
    B = tab.params.shape[0]
    P = np.ascontiguousarray(tab.params)
    pi, pf = P.view(np.int32), P.view(np.float32)
    full = int(bool(np.any(P[:, abi.CLP_FLAGS] & (abi.CLF_THERMAL | abi.CLF_OUTAGE | abi.CLF_DYNAMICS))))
    state = np.zeros((B, 8), dtype=np.float32)
    d64 = P[:, abi.CLP_D_FIRST:abi.CLP_D_LAST + 1].copy().view(np.float64)
    state[:, 0], state[:, 1], state[:, 2] = pf[:, abi.CLP_L_SOC0], pf[:, abi.CLP_L_EFF0], pf[:, abi.CLP_L_CAP]
    state[:, 3], state[:, 4], state[:, 5] = pf[:, abi.CLP_CS_SOC0], pf[:, abi.CLP_HS_SOC0], pf[:, abi.CLP_DS_SOC0]
    state[:, 6] = d64[:, abi.CLPD_EFF0] - state[:, 1].astype(np.float64)              # what cl_reset_kernel writes
    state[:, 7] = d64[:, abi.CLPD_CAP] - state[:, 2].astype(np.float64)
    cap32 = pf[:, abi.CLP_L_CAP].copy()
    if f64 == 2:
        state[:, 2] = 0.0                                                               # what cl_reset_kernel writes under CLD_F64_CHAIN
    mism = 0
    has_batt = (P[:, abi.CLP_FLAGS] & abi.CLF_BATTERY) != 0
    acts = g.ref['actions']
    worst = {}
    out, rw = np.zeros(10, dtype=np.float32), np.zeros(1, dtype=np.float32)
    vp = ctypes.c_void_p
    for t in range(g.facts['steps']):
        for b in range(B):
            col = lambda slot: float(acts[t][pi[b, slot]]) if pi[b, slot] >= 0 else 0.0
            a = [col(abi.CLP_ACT_COOL_STO), col(abi.CLP_ACT_HEAT_STO), col(abi.CLP_ACT_DHW_STO), col(abi.CLP_ACT_ELEC_STO),
                 col(abi.CLP_ACT_COOL_DEV), col(abi.CLP_ACT_HEAT_DEV)]
            if pi[b, abi.CLP_ACT_COH_DEV] >= 0:
                c = col(abi.CLP_ACT_COH_DEV)
                a[4], a[5] = abs(min(c, 0.0)), abs(max(c, 0.0))
            a6 = np.asarray(a, dtype=np.float32)
            row = np.ascontiguousarray(tab.ts[t, b])
            st = state[b]
            lib.host_unit_step(P[b].ctypes.data_as(vp), row.ctypes.data_as(vp), t, 1, 0, full, int(f64), a6.ctypes.data_as(vp),
                               st.ctypes.data_as(vp), out.ctypes.data_as(vp), rw.ctypes.data_as(vp))
            degcap = cap32[b] - st[2] if f64 == 2 else st[2]
            mism += bool(has_batt[b] and st[0] != np.float32(g.ref['soc'][t][b]))
            for key, got in (('soc', st[0]), ('eff', st[1]), ('degcap', degcap), ('cs_soc', st[3]), ('hs_soc', st[4]), ('ds_soc', st[5]), ('net', out[0])):
                if key in ('soc', 'eff', 'degcap') and not has_batt[b]:
                    continue
                ref = float(g.ref[key][t][b])
                worst[key] = max(worst.get(key, 0.0), abs(float(got) - ref) / (1e-4 + 1e-4 * abs(ref)))
    worst['soc_mismatch_fraction'] = mism / max(1, int(has_batt.sum()) * g.facts['steps'])
    return worst


@pytest.mark.parametrize('name', ['g2022_all', 'g2020_cz1', 'g2023_p2', 'g2020_15min', 'g2023_heat'])
def test_free_running_f64_unit_stays_on_the_reference_trajectory(shim, name):
    worst = free_run(shim, name, True)
    assert worst.pop('soc_mismatch_fraction') == 0.0
    assert max(worst.values()) < 0.1, worst
    # the battery state: the reference's own float32 values, bit for bit, after a whole free-running episode
    assert worst['soc'] == 0.0 and worst['eff'] == 0.0 and worst['degcap'] == 0.0, worst


def test_fp32_unit_drifts_on_the_expansive_part_of_the_battery_map(shim):
    """What CLD_F64_MAPS is for: the same free run with the fp32 map leaves the 1e-4 bar on the 2020 fixture (and stays inside 1e-3)."""
    worst = free_run(shim, 'g2020_cz1', False)
    assert worst.pop('soc_mismatch_fraction') > 0.5                  # hardly a step ends on the reference's float32 soc
    assert 1.0 < max(worst.values()) < 10.0, worst


@pytest.mark.parametrize('name', ['g2022_all', 'g2020_cz1', 'g2023_p2', 'g2020_15min', 'g2023_heat', 's_2021', 's_2020_cz3', 's_2023_p3', 's_baeda'])
def test_free_running_f64_chain_unit_reaches_the_north_star_bar(shim, name):
    """CLD_F64_CHAIN (`cl::battery_charge_chain`): the soc chain in float64 and the degraded capacity carried as the loss
    `capacity - degraded_capacity` in its float32 plane -- the default three state planes, no float64 division, no segment selection.
    Free-running over whole fixtures, outage rows included: every quantity inside 0.2 x (1e-4 + 1e-4 |ref|) -- the fp32 map sits at
    2.1 x on the 2020 fixture's net -- and the battery ends a step on the reference's own float32 soc in ~9 of 10 steps."""
    worst = free_run(shim, name, 2)
    frac = worst.pop('soc_mismatch_fraction')
    assert max(worst.values()) < 0.2, worst
    assert frac < 0.25, frac


def test_markstein_division_returns_the_ieee_quotient(shim):
    """`cl::div_rn` (round 4: the float64 battery map divides by its constant divisors through their correctly rounded reciprocals and two
    fused-multiply-add correction steps): the result must be the correctly rounded quotient -- what the reference's division returns --
    for every operand pair, not just the fixtures': four million pairs over eleven decades, divisors with all-ones / near-power-of-two
    significands (the classic hard cases of reciprocal-based division), zero and negative numerators; a non-finite reciprocal takes the
    plain division."""
    shim.host_div_rn_mismatches.restype = ctypes.c_long
    shim.host_div_rn_mismatches.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long]
    rng = np.random.RandomState(11)
    n = 1 << 22
    a = rng.uniform(-1, 1, n) * 10.0 ** rng.uniform(-6, 5, n)
    b = rng.uniform(0.5, 2, n) * 10.0 ** rng.uniform(-6, 5, n)
    hard = np.concatenate([np.nextafter(2.0 ** rng.randint(-20, 20, 4096), 0), np.nextafter(2.0 ** rng.randint(-20, 20, 4096), np.inf),
                           2.0 ** rng.randint(-20, 20, 4096).astype(float), np.float64(1e-6) * np.ones(16)])
    b[:hard.size] = hard
    a[hard.size:hard.size + 1024] = 0.0
    assert shim.host_div_rn_mismatches(a.ctypes.data, b.ctypes.data, n) == 0
    # ... and the curve-segment form the map uses: (y1 - y0) (x - x0) / (x1 - x0) with the shipped battery curves' breakpoints
    xs = np.array([0.0, 0.3, 0.7, 0.8, 1.0])
    num = rng.uniform(-1, 1, 1 << 16) * rng.uniform(0, 1, 1 << 16)
    for d in np.diff(xs):
        den = np.full(num.size, d)
        assert shim.host_div_rn_mismatches(num.ctypes.data, den.ctypes.data, num.size) == 0
    z = np.zeros(4)
    assert shim.host_div_rn_mismatches(np.ones(4).ctypes.data, z.ctypes.data, 4) == 0          # 1 / 0: inf both ways


def _check_step(lib, P, row, t, f64, a6, st, full=1):
    vp = ctypes.c_void_p
    lib.host_unit_step_check.restype = ctypes.c_uint
    lib.host_unit_step_check.argtypes = [vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp]
    row = np.ascontiguousarray(row, dtype=np.float32)
    return int(lib.host_unit_step_check(P.ctypes.data_as(vp), row.ctypes.data_as(vp), t, 1, full, f64, a6.ctypes.data_as(vp), st.ctypes.data_as(vp)))


@pytest.mark.parametrize('f64', [0, 1, 2])
def test_check_unit_reports_the_references_assertions(shim, f64):
    """CLD_CHECK on the CPU (the device header compiled with g++): `cl::unit_step<.., CHECK = true>` returns no bit on valid rows of the
    reference-run fixtures, and exactly the bit of the assertion the reference would raise on a corrupted row -- a negative non-shiftable load
    (energy_model.py:146-148), a negative cooling demand (building.py:1660, 1831-1835), an outage at t = 0 where reset() has already booked
    the ideal loads (building.py:665)."""
    g = golden('g2020_cz1')
    tab = g.spec().episode_tables(0)
    P = np.ascontiguousarray(tab.params)
    pf = P.view(np.float32)
    b = 4
    st0 = np.zeros(8, dtype=np.float32)
    st0[0], st0[1], st0[2] = pf[b, abi.CLP_L_SOC0], pf[b, abi.CLP_L_EFF0], (0.0 if f64 == 2 else pf[b, abi.CLP_L_CAP])
    a6 = np.zeros(6, dtype=np.float32)
    for t in range(0, 40):
        assert _check_step(shim, P[b], tab.ts[t, b], t, f64, a6, st0.copy()) == 0, t
    row = tab.ts[7, b].copy(); row[abi.CLT_NSL] = -1.0
    assert _check_step(shim, P[b], row, 7, f64, a6, st0.copy()) == abi.CLV_NSL
    row = tab.ts[7, b].copy(); row[abi.CLT_COOL_DEM] = -3.0
    assert _check_step(shim, P[b], row, 7, f64, a6, st0.copy()) & abi.CLV_COOLING
    Pb = P[b].copy()
    for slot in (abi.CLP_FLAGS, abi.CLP_L_FLAGS, abi.CLP_F_FLAGS):          # (the flag word and its copies in the lean / thermal blocks)
        Pb[slot] |= abi.CLF_OUTAGE
    row = tab.ts[0, b].copy(); row[abi.CLT_OUTAGE] = 1.0
    assert _check_step(shim, Pb, row, 0, f64, a6, st0.copy()) & abi.CLV_FLEXIBILITY
    assert _check_step(shim, Pb, row, 5, f64, a6, st0.copy()) == 0          # the same outage later in the episode: nothing pre-booked, flexibility = |solar| >= 0
    # the lean unit (battery + PV + load): only the load's polarity can trip
    g2 = golden('g2022_all')
    tab2 = g2.spec().episode_tables(0)
    P2 = np.ascontiguousarray(tab2.params)
    row = tab2.ts[3, 1].copy()
    assert _check_step(shim, P2[1], row, 3, f64, a6, st0.copy(), full=0) == 0
    row[abi.CLT_NSL] = -0.5
    assert _check_step(shim, P2[1], row, 3, f64, a6, st0.copy(), full=0) == abi.CLV_NSL


def test_device_header_under_address_and_undefined_behaviour_sanitizers(tmp_path):
    """SURVEY section 5 / VERDICT r05 item 9: the device's unit header (`csrc/cl_unit.h`, every precision model, the CHECK variant) built with
    `-fsanitize=address,undefined -fno-sanitize-recover=all` and free-run over fixtures in a child process (the sanitizer runtime has to be
    loaded before the interpreter's first allocation: LD_PRELOAD).  GPU AddressSanitizer is not available on this pool; the CPU build of the
    same arithmetic is what can be sanitized.  Any report aborts the child."""
    import os
    import sys
    so = tmp_path / 'libcl_unit_host_san.so'
    subprocess.run(['g++', '-O1', '-g', '-shared', '-fPIC', '-DCL_HOST_SHIM', '-ffp-contract=off', '-fsanitize=address,undefined', '-fno-sanitize-recover=all',
                    str(HERE / 'cl_unit_host.cpp'), '-o', str(so)], check=True)
    asan = subprocess.run(['gcc', '-print-file-name=libasan.so'], capture_output=True, text=True, check=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip('no libasan runtime beside gcc')
    code = f'''
import ctypes, sys
sys.path.insert(0, {str(Path(__file__).resolve().parent)!r}); sys.path.insert(0, {str(Path(__file__).resolve().parent.parent)!r})
import numpy as np
import test_f64_maps_host as T
lib = ctypes.CDLL({str(so)!r})
vp = ctypes.c_void_p
lib.host_unit_step.argtypes = [vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp, vp, vp]
for name in ('g2023_p2', 's_2023_p3'):
    for f64 in (0, 1, 2):
        w = T.free_run(lib, name, f64)
        w.pop('soc_mismatch_fraction')
        assert max(w.values()) < 10.0, (name, f64, w)
from golden_util import golden
from citylearn_amd import abi
g = golden('g2023_p2'); tab = g.spec().episode_tables(0); P = np.ascontiguousarray(tab.params)
a6 = np.zeros(6, dtype=np.float32)
for t in range(380, 410):
    for b in range(P.shape[0]):
        for f64 in (0, 1, 2):
            st = np.zeros(8, dtype=np.float32); st[0] = 0.4; st[1] = 0.9; st[2] = 0.0 if f64 == 2 else P.view(np.float32)[b, abi.CLP_L_CAP]
            assert T._check_step(lib, P[b], tab.ts[t, b], t, f64, a6, st) == 0
print('sanitized run ok')
'''
    env = {**os.environ, 'LD_PRELOAD': asan, 'ASAN_OPTIONS': 'detect_leaks=0:abort_on_error=1', 'UBSAN_OPTIONS': 'halt_on_error=1:print_stacktrace=1'}
    p = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0 and 'sanitized run ok' in p.stdout, (p.stdout[-2000:], p.stderr[-4000:])
