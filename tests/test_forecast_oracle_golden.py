"""Pins oracle/forecast_oracle.py to the reference (tests/golden/forecast.npz).  CPU only."""
import numpy as np

from conftest import load_golden
from oracle import forecast_oracle as fo


def test_marginal_forecast_matches_cython():
    g, p = load_golden('forecast.npz'), load_golden('post.npz')
    S = 12
    got = fo.marginal_forecast(g['mf_x'], p['u_Xs'][:S, -1], p['u_zs'][:S, -1],
                               p['u_weights'][:S, -1], p['u_mus'][:S], p['u_sigmas'][:S],
                               p['u_intercepts'][:S].ravel(), p['u_lambdas'][:S].ravel())
    np.testing.assert_allclose(got, g['mf_probas'], rtol=1e-12, atol=1e-15)


def test_mean_probas_is_the_map_forecast_for_one_sample():
    g, p = load_golden('forecast.npz'), load_golden('post.npz')
    # forecast_probas_map_ = expit(b - dist(X_ahead)): recover X_ahead's distances from it
    b = p['u_intercepts'][int(g['best'])][0]
    d = b - np.log(g['map'] / (1 - g['map']))
    assert np.allclose(np.diag(d), 0.0, atol=1e-12) and np.allclose(d, d.T)
