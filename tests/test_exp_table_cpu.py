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


def _table(name='c_exp2_tab['):
    text = open(HDR).read()
    at = text.index(name)
    body = text[text.index('{', at) + 1:text.index('};', at)]
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


def test_table11_entries_are_correctly_rounded():
    """c_exp2_tab11: 2^(j / 2048), the table of tab_exp11 (the pipelined sweeps' evaluators)"""
    tab = _table('c_exp2_tab11[')
    assert tab.shape == (2048,)
    getcontext().prec = 60
    ln2 = Decimal(2).ln()
    want = np.array([float((ln2 * j / 2048).exp()) for j in range(2048)])
    np.testing.assert_array_equal(tab, want)
    assert tab[0] == 1.0 and np.all(np.diff(tab) > 0) and tab[-1] < 2.0
    np.testing.assert_array_equal(tab[::8], _table())           # the 256-entry table is every eighth entry


def test_table11_exponential_degree_three_error_bound():
    """tab_exp11 (device_common.hpp) restated: one reduction step with ln2 / 2048 rounded to double
    (exact product inside the fma: emulated in longdouble), degree-3 polynomial on |r| <= ln2 / 4096.
    Same bound as the 256-entry form: 2 ulp + |x| / 2 ulp of the correctly rounded exponential."""
    tab = _table('c_exp2_tab11[')
    getcontext().prec = 50
    magic = 6755399441055744.0
    c = float.fromhex('0x1.62e42fefa39efp-12')
    assert c == 0.6931471805599453 / 2048
    x = -np.random.RandomState(2).uniform(0.0, 80.0, 20000)
    t = x * 2954.639443740597 + magic
    kf = t - magic
    ki = (t.view(np.int64) & 0xFFFFFFFF).astype(np.int64)
    ki = np.where(ki >= 2 ** 31, ki - 2 ** 32, ki)
    assert np.array_equal(ki, kf.astype(np.int64))              # the low word IS the integer
    r = (x.astype(np.longdouble) - kf.astype(np.longdouble) * np.longdouble(c)).astype(np.float64)
    assert np.max(np.abs(r)) <= 0.6931471805599453 / 4096 * (1 + 1e-6) + 80 * 1.2e-16
    p = r * (1.0 / 6.0) + 0.5
    p = p * r + 1.0
    p = p * r + 1.0
    got = np.ldexp(tab[ki & 2047] * p, (ki >> 11).astype(np.int32))
    want = np.array([float(Decimal(float(v)).exp()) for v in x])
    err = np.abs(got - want) / np.spacing(want)
    assert np.all(err <= 2.0 + 0.5 * np.abs(x))
    assert np.max(err[np.abs(x) < 10.0]) <= 7.0
    # truncation alone (the next Taylor term) stays below a third of an ulp
    assert (0.6931471805599453 / 4096) ** 4 / 24 < 0.35 * 2.0 ** -53
