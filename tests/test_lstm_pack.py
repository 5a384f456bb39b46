"""Host-side packing of the LSTM weights for the matrix-core kernel (no GPU): the split 16-bit A-operand fragments."""
import numpy as np
import pytest

from citylearn_amd import abi
from citylearn_amd.dynamics import WHH0, WIH1, WHH1, pack_lstm_split


def _fragment(lstm_w, base, rb):
    """[B, 64 lanes, 8]: W[32 rb + (lane & 31)][unit (j & 3) + 8 (j >> 2) + 4 (lane >> 5)] (csrc/cl_lstm.h)."""
    lane, j = np.arange(64), np.arange(8)
    unit = (j[None, :] & 3) + 8 * (j[None, :] >> 2) + 4 * (lane[:, None] >> 5)
    Wm = lstm_w[:, base:base + 1024].reshape(lstm_w.shape[0], 64, 16)
    return Wm[:, (32 * rb + (lane & 31))[:, None], unit]


@pytest.mark.parametrize('fmt,terms,bound', [('bf16', 3, 2.0 ** -24), ('f16', 2, 2.0 ** -22)])
def test_split_fragments_reproduce_the_weights(fmt, terms, bound):
    """The terms of every fragment add up to the fp32 weight: exactly for three bf16 terms, within 2^-22 relative (2^-25
    absolute where the second term is subnormal) for two f16 terms; the block keeps the CL_LSTM_NWB stride of the C-ABI."""
    rng = np.random.RandomState(3)
    w = (rng.randn(3, abi.CL_LSTM_NW) * rng.choice([1e-3, 0.1, 1.0, 8.0], size=(3, abi.CL_LSTM_NW))).astype(np.float32)
    out = pack_lstm_split(w, fmt)
    assert out.shape == (3, 18, 64, 8) and out.dtype == np.uint16 and out[0].size == abi.CL_LSTM_NWB
    if fmt == 'f16':
        assert not out[:, 12:].any()                       # two terms: the first 6144 words of every block (CLD_LSTM_F16)
    for m, base in enumerate((WHH0, WIH1, WHH1)):
        for rb in range(2):
            want = _fragment(w, base, rb).astype(np.float64)
            got = np.zeros_like(want)
            for k in range(terms):
                bits = out[:, (m * 2 + rb) * terms + k]
                term = (bits.astype(np.uint32) << 16).view(np.float32) if fmt == 'bf16' else bits.view(np.float16)
                got += term.astype(np.float64)
            err = np.abs(got - want)
            if fmt == 'bf16':
                assert (err <= bound * np.abs(want)).all()
            else:
                assert (err <= np.maximum(bound * np.abs(want), 2.0 ** -25)).all()


def test_f16_split_refuses_weights_outside_the_f16_range():
    w = np.zeros((1, abi.CL_LSTM_NW), dtype=np.float32)
    w[0, WHH0] = 7.0e4
    with pytest.raises(ValueError):
        pack_lstm_split(w, 'f16')
    pack_lstm_split(w, 'bf16')
