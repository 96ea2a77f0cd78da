"""Host-side convergence diagnostics (dynetlsm_amd.diagnostics; trace_utils.py of the reference).
No reference fixture exists for these (the reference needs statsmodels for its Yule-Walker solve,
which is not installed where the goldens were made): known-answer checks only."""
import numpy as np
import pytest

from dynetlsm_amd import diagnostics as dg


def _ar2(n, phi, seed):
    rng = np.random.RandomState(seed)
    e = rng.randn(n)
    x = np.zeros(n)
    for t in range(2, n):
        x[t] = phi[0] * x[t - 1] + phi[1] * x[t - 2] + e[t]
    return x


def test_yule_walker_solves_the_adjusted_normal_equations():
    x = _ar2(3000, (0.5, -0.3), 0)
    d = x - x.mean()
    n, p = d.shape[0], 4
    r = np.array([np.dot(d[:n - k], d[k:]) / (n - k) for k in range(p + 1)])
    R = np.array([[r[abs(i - j)] for j in range(p)] for i in range(p)])
    rho, sigma = dg.yule_walker_adjusted(x, p)
    np.testing.assert_allclose(rho, np.linalg.solve(R, r[1:]), rtol=1e-10)
    np.testing.assert_allclose(sigma ** 2, r[0] - r[1:].dot(rho), rtol=1e-10)


def test_ar_fit_recovers_a_known_process():
    phi = (0.5, -0.3)
    x = _ar2(200000, phi, 1)
    rho, sigma = dg.yule_walker_adjusted(x, 2)
    np.testing.assert_allclose(rho, phi, atol=0.01)
    np.testing.assert_allclose(sigma, 1.0, atol=0.01)
    var0, order = dg.spectrum0_ar(x[:20000])
    assert order >= 2
    np.testing.assert_allclose(var0 * 20000, 1.0 / (1 - sum(phi)) ** 2, rtol=0.15)
    assert dg.spectrum0_ar(np.ones(50)) == (0., 0.)


def test_geweke_separates_stationary_from_drifting_traces():
    x = _ar2(4000, (0.5, -0.3), 2)
    z, p = dg.geweke_diag(x)
    assert abs(z) < 3 and 0 <= p <= 1
    drift = x + np.linspace(0, 5, x.shape[0])
    zd, pd_ = dg.geweke_diag(drift)
    assert abs(zd) > 5 and pd_ < 1e-6
    # burn-in removes an initial transient
    burnt = np.concatenate([np.full(1000, 50.0), x])
    zb, _ = dg.geweke_diag(burnt, n_burn=1000)
    np.testing.assert_allclose(zb, z)


def test_effective_n():
    rng = np.random.RandomState(3)
    white = rng.randn(20000)
    assert 0.8 * 20000 < dg.effective_n(white) < 1.25 * 20000
    x = _ar2(20000, (0.9, 0.0), 4)                   # AR(1): n (1 - phi) / (1 + phi)
    assert 700 < dg.effective_n(x, maxlags=50) < 1600          # theory 1058
    with pytest.raises(ValueError):
        dg.effective_n(white[:50], maxlags=100)
