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
