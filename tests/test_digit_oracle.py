"""CPU checks of the integer restatement of the int8-digit dense pass (oracle/tramp_oracle.py: digit_split, digit_matvec;
device: tramp_amd/csrc/gemm_sliced.h).  The GPU side compares bit for bit with it (tests/test_gpu_matvec_digits.py)."""
import numpy as np
import pytest

from oracle import tramp_oracle as orc


@pytest.mark.parametrize("S", [4, 6, 7, 8])
def test_digits_reconstruct_the_operand(S):
    rs = np.random.RandomState(S)
    a = rs.standard_normal((40, 333)) * 10.0 ** rs.uniform(-6, 6, (40, 1))
    a[3] = 0.0
    a[5, 7] = 1.0                                   # an exact power of two as the row maximum
    a[5, :7] *= 1e-3
    a[5, 8:] *= 1e-3
    dig, sc = orc.digit_split(a, S)
    assert dig.min() >= -128 and dig.max() <= 127 and np.abs(dig[0]).max() <= 65
    assert np.all(sc[np.arange(40) != 3] >= np.abs(a).max(axis=1)[np.arange(40) != 3]) and sc[3] == 1.0
    assert np.all(np.log2(sc) == np.round(np.log2(sc)))
    w = 2.0 ** -(6 + 8 * np.arange(S))
    back = np.zeros_like(a, dtype=np.longdouble)
    for i in range(S):
        back += dig[i].astype(np.longdouble) * np.longdouble(w[i])
    back *= sc[:, None]
    # N = rint(a 2^(6 + 8 (S - 1)) / scale): half a unit of the last digit
    assert np.all(np.abs(back - a) <= 0.5 * sc[:, None] * 2.0 ** -(6 + 8 * (S - 1)) * (1 + 1e-12))
    if S >= 7:                                      # fixed point against the row maximum: entries within 2x of it are exact
        near = np.abs(a) * 2 >= np.abs(a).max(axis=1, keepdims=True)
        assert np.array_equal(back.astype(np.float64)[near], a[near])


@pytest.mark.parametrize("S", [6, 7, 8])
@pytest.mark.parametrize("shape", [(20, 64), (33, 700), (5, 20001)])
def test_digit_product_within_its_bound(S, shape):
    rs = np.random.RandomState(S + shape[1])
    mat = rs.standard_normal(shape) / np.sqrt(shape[1])
    x = rs.standard_normal((7, shape[1])) * np.array([1, 40, 0.1, 3, 7, 1e-3, 2])[:, None]
    x[:, ::5] *= 1e-3
    out = orc.digit_matvec(mat, x, S)
    ref = x.astype(np.longdouble) @ mat.astype(np.longdouble).T
    k = shape[1]
    amax, asum = np.abs(mat).max(axis=1)[None, :], np.abs(mat).sum(axis=1)[None, :]
    xmax, xsum = np.abs(x).max(axis=1)[:, None], np.abs(x).sum(axis=1)[:, None]
    u = 2.0 ** (2 - 8 * S)
    bound = u * (4 * (S - 1) * k * amax * xmax + amax * xsum + xmax * asum) + 2.0 ** -50 * np.abs(ref).astype(np.float64)
    assert np.all(np.abs(out - ref) <= bound)
    if S == 8:
        assert np.abs(out - ref).max() <= np.abs(x @ mat.T - ref).max() + 1e-300


def test_special_rows():
    mat = np.arange(12.0).reshape(3, 4) - 5
    x = np.ones((4, 4))
    x[1, 2] = np.nan
    x[2, 0] = np.inf
    x[3] = 0
    out = orc.digit_matvec(mat, x, 7)
    assert np.all(np.isnan(out[1])) and np.all(np.isnan(out[2])) and np.all(out[3] == 0)
    assert np.array_equal(out[0], mat.sum(axis=1))
    assert orc.digit_scale(0.0) == 1.0 and np.isnan(orc.digit_scale(np.inf)) and orc.digit_scale(1.0) == 2.0
    assert orc.digit_scale(0.75) == 1.0 and orc.digit_scale(1e-320) == 2.0 ** -960 and orc.digit_scale(2.0 ** -970) == 2.0 ** -960
    assert np.isnan(orc.digit_scale(2.0 ** 1023)) and orc.digit_scale(2.0 ** 1022) == 2.0 ** 1023
