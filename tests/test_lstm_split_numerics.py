"""The split-operand arithmetic of the LSTM kernel restated in numpy (no GPU): what the 16-bit matrix cores compute from the
two-term f16 / three-term bf16 operands, against the exact product.  Pins the error bounds quoted in csrc/cl_lstm.h and DESIGN.md."""
import numpy as np
import pytest

from citylearn_amd.dynamics import _bf16_split3, _f16_split2


def _terms(x, fmt):
    """fp32 array -> list of float64 arrays, the 16-bit terms the kernel feeds to the matrix cores (round to nearest even)."""
    if fmt == 'f16':
        return [t.view(np.float16).astype(np.float64) for t in _f16_split2(x)]
    return [(t.astype(np.uint32) << 16).view(np.float32).astype(np.float64) for t in _bf16_split3(x)]


@pytest.mark.parametrize('fmt,pairs,rel', [('f16', [(1, 0), (0, 1), (0, 0)], 3 * 2.0 ** -22),
                                           ('bf16', [(2, 0), (1, 1), (0, 2), (1, 0), (0, 1), (0, 0)], 3 * 2.0 ** -24)])
def test_partial_products_reproduce_the_fp32_product(fmt, pairs, rel):
    """sum over the kernel's partial products A_i B_j (exact in the fp32 accumulator's input: 11 x 11 / 8 x 8 bit mantissas) vs W h
    in float64: within `rel` * sum |W||h| (+ the 2^-25 absolute floor per f16 operand whose second term is subnormal)."""
    rng = np.random.RandomState(11)
    for scale in (1.0, 0.05, 4.0):
        W = (rng.randn(64, 16) * scale).astype(np.float32)              # gate rows x hidden units
        h = np.tanh(rng.randn(16, 32) * 1.5).astype(np.float32) * rng.uniform(0.0, 1.0, size=(16, 32)).astype(np.float32)   # |h| < 1
        Wt, ht = _terms(W, fmt), _terms(h, fmt)
        got = sum(Wt[i] @ ht[j] for i, j in pairs)
        want = W.astype(np.float64) @ h.astype(np.float64)
        bound = rel * (np.abs(W).astype(np.float64) @ np.abs(h).astype(np.float64))
        if fmt == 'f16':
            bound = bound + 2.0 ** -25 * (np.abs(W).sum(axis=1, keepdims=True) + np.abs(h).sum(axis=0, keepdims=True))
        assert (np.abs(got - want) <= bound).all(), float((np.abs(got - want) / bound).max())


def test_device_side_split_of_the_hidden_state_is_exact_enough():
    """h = h0 + h1 + r with the kernel's round-to-nearest conversions: |r| <= 2^-22 |h| or 2^-25 absolute (subnormal second term);
    the three bf16 terms reproduce h bit for bit."""
    rng = np.random.RandomState(5)
    h = (np.tanh(rng.randn(4096)) * 10.0 ** rng.uniform(-6, 0, size=4096)).astype(np.float32)
    t = _terms(h, 'f16')
    r = np.abs(h.astype(np.float64) - t[0] - t[1])
    assert (r <= np.maximum(2.0 ** -22 * np.abs(h), 2.0 ** -25)).all()
    t = _terms(h, 'bf16')
    assert np.array_equal((t[0] + t[1] + t[2]).astype(np.float32), h)


def test_common_denominator_cell_update_matches_the_plain_one():
    """The experimental cell update of cl_lstm.h (lstm_variant 32: 7 instead of 10 transcendentals per unit and cell) restated in
    float32 numpy against the plain formulation, on pre-scaled gate values z (2^z = exp(-x)) up to the admitted bound
    z_i + z_f + z_g < 126, z_o < 62, with cell states of both signs and sizes: same values to a few fp32 roundings, always finite."""
    rng = np.random.RandomState(2)
    f32 = np.float32
    n = 200000
    z = (rng.uniform(-40.0, 40.0, size=(4, n))).astype(f32)
    z[:, : n // 4] = rng.uniform(-3.0, 3.0, size=(4, n // 4)).astype(f32)              # the typical range
    c = (rng.randn(n) * 10.0 ** rng.uniform(-3, 1.5, size=n)).astype(f32)
    one, two = f32(1.0), f32(2.0)
    with np.errstate(over='ignore'):
        ei, ef, eg, eo = (np.exp2(z[k]) for k in range(4))
        # plain: lstm_act
        gi, gf, gg, go = one / (one + ei), one / (one + ef), two / (one + eg) - one, one / (one + eo)
        c_ref = gf * c + gi * gg
        h_ref = go * (two / (one + np.exp2(c_ref * f32(-2.885390043258667))) - one)
        # common denominators
        pf, t = one + ef, (one + ei) * (one + eg)
        c_new = (c * t + (one - eg) * pf) * (one / (pf * t))
        ec = np.exp2(np.minimum(c_new * f32(-2.885390043258667), f32(64.0)))
        h_new = (one - ec) * (one / ((one + eo) * (one + ec)))
    assert np.isfinite(c_new).all() and np.isfinite(h_new).all()
    assert np.abs(c_new - c_ref).max() <= 1e-6 * (1.0 + np.abs(c_ref).max())
    assert (np.abs(c_new - c_ref) <= 4e-7 * (1.0 + np.abs(c_ref))).all()
    assert np.abs(h_new - h_ref).max() <= 1e-6
