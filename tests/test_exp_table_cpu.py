"""The table of the engine's table-based exponential (dynetlsm_amd/csrc/exp2_table.hpp, used by
tab_exp in device_common.hpp): every entry is 2^(j/256) correctly rounded, and the algorithm the
kernels run on it - restated here in numpy without fused multiply-adds - stays within 2 ulp of
the correctly rounded exponential over the range the sweeps use."""
import os
import re
from decimal import Decimal, getcontext

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, 'dynetlsm_amd', 'csrc', 'exp2_table.hpp')


def _table():
    text = open(HDR).read()
    body = text[text.index('{', text.index('c_exp2_tab')) + 1:text.index('};')]
    vals = [float.fromhex(tok) for tok in re.findall(r'0x[0-9a-fA-F.]+p[+-]?\d+', body)]
    return np.array(vals)


def test_table_entries_are_correctly_rounded():
    tab = _table()
    assert tab.shape == (256,)
    getcontext().prec = 60
    ln2 = Decimal(2).ln()
    want = np.array([float((ln2 * j / 256).exp()) for j in range(256)])
    np.testing.assert_array_equal(tab, want)
    assert tab[0] == 1.0 and np.all(np.diff(tab) > 0) and tab[-1] < 2.0


def test_table_exponential_one_step_reduction_error_grows_with_the_argument():
    """the kernels' default: ONE reduction step with ln2 / 256 rounded to double (the product is exact
    inside the fma, here emulated in longdouble) - the constant's rounding error, |x| 1.1e-16, is the
    whole price: within 2 ulp + |x| / 2 ulp of the correctly rounded exponential"""
    tab = _table()
    getcontext().prec = 50
    magic = 6755399441055744.0
    c = float.fromhex('0x1.62e42fefa39efp-9')
    assert c == 0.6931471805599453 / 256
    x = -np.random.RandomState(1).uniform(0.0, 80.0, 20000)
    t = x * 369.3299304675746 + magic
    kf = t - magic
    ki = kf.astype(np.int64)
    r = (x.astype(np.longdouble) - kf.astype(np.longdouble) * np.longdouble(c)).astype(np.float64)
    p = r * (1.0 / 24.0) + 1.0 / 6.0
    p = p * r + 0.5
    p = p * r + 1.0
    p = p * r + 1.0
    got = np.ldexp(tab[ki & 255] * p, (ki >> 8).astype(np.int32))
    want = np.array([float(Decimal(float(v)).exp()) for v in x])
    err = np.abs(got - want) / np.spacing(want)
    assert np.all(err <= 2.0 + 0.5 * np.abs(x))
    near = np.abs(x) < 10.0
    assert np.max(err[near]) <= 7.0


def test_table_exponential_within_two_ulp():
    tab = _table()
    getcontext().prec = 50
    magic = 6755399441055744.0                                   # 1.5 * 2^52
    x = -np.random.RandomState(0).uniform(0.0, 80.0, 20000)
    t = x * 369.3299304675746 + magic
    kf = t - magic
    ki = (t.view(np.int64) & 0xFFFFFFFF).astype(np.int64)
    ki = np.where(ki >= 2 ** 31, ki - 2 ** 32, ki)
    assert np.array_equal(ki, kf.astype(np.int64))              # the low word IS the integer
    r = x - kf * float.fromhex('0x1.62e42fee00000p-9')
    r = r - kf * float.fromhex('0x1.a39ef35793c76p-41')
    assert np.max(np.abs(r)) <= 0.6931471805599453 / 512 * (1 + 1e-9)
    p = r * (1.0 / 24.0) + 1.0 / 6.0
    p = p * r + 0.5
    p = p * r + 1.0
    p = p * r + 1.0
    got = np.ldexp(tab[ki & 255] * p, (ki >> 8).astype(np.int32))
    want = np.array([float(Decimal(float(v)).exp()) for v in x])
    ulp = np.spacing(want)
    assert np.max(np.abs(got - want) / ulp) <= 2.0
