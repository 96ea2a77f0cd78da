"""Pins oracle/post_oracle.py (post-loop processing) to the reference's
label_utils / model_selection functions (tests/golden/post.npz)."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import oracle as orc
from oracle import post_oracle as po


@pytest.fixture(scope='module')
def g():
    return load_golden('post.npz')


@pytest.mark.parametrize('tag', ['u', 'd'])
def test_cooccurrence_vi_and_selection(g, tag):
    zs, n_burn, K = g[tag + '_zs'], int(g[tag + '_n_burn']), int(g[tag + '_K'])
    cooc = po.posterior_cooccurrence(zs, n_burn, K)
    np.testing.assert_allclose(cooc, g[tag + '_cooc'], rtol=1e-14, atol=1e-15)
    Y, Xs, ics = g[tag + '_Y'], g[tag + '_Xs'], g[tag + '_intercepts']

    def loglik(i):
        if tag == 'd':
            return orc.dynamic_network_loglikelihood_directed(Y, Xs[i], ics[i, 0], ics[i, 1],
                                                              g['d_radiis'][i])
        return orc.dynamic_network_loglikelihood_undirected(Y, Xs[i], ics[i, 0])
    best, vis = po.minimize_expected_vi(zs, n_burn, cooc, loglik)
    np.testing.assert_allclose(vis, g[tag + '_vis'], rtol=1e-12)
    assert best == int(g[tag + '_best'])
    np.testing.assert_array_equal(po.cluster_counts(zs, n_burn), g[tag + '_counts'])


def test_identical_partitions_tie_exactly(g):
    zs, n_burn = g['d_zs'], int(g['d_n_burn'])
    vis = g['d_vis']
    assert vis[12 - n_burn] == vis[20 - n_burn] == vis[33 - n_burn]
