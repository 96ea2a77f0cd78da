"""a7 (case_control_likelihood.py:45-68): the product's host-side edge-table builder
against the tables the reference's DirectedCaseControlSampler.init produced
(tests/golden/likelihoods.npz, recorded by make_golden.py).  SURVEY.md 8c: exactly."""
import numpy as np
import pytest

from conftest import load_golden


@pytest.mark.parametrize('tag', ['a', 'b', 'c'])
def test_build_edge_lists_matches_reference(tag):
    from dynetlsm_amd.case_control import build_edge_lists
    g = load_golden('likelihoods.npz')
    Y = g[tag + '_Yd']
    deg, ie, oe = build_edge_lists(Y)
    assert deg.dtype == np.int64 and ie.dtype == np.int64 and oe.dtype == np.int64
    np.testing.assert_array_equal(deg, g[tag + '_degrees'])
    np.testing.assert_array_equal(ie, g[tag + '_in_edges'])
    np.testing.assert_array_equal(oe, g[tag + '_out_edges'])


def test_build_edge_lists_corner_cases():
    from dynetlsm_amd.case_control import build_edge_lists
    # an empty slice, an isolated node, a full row / column
    Y = np.zeros((3, 5, 5))
    Y[1, 0, 1:] = 1
    Y[2, 1:, 0] = 1
    deg, ie, oe = build_edge_lists(Y)
    assert ie.shape == (3, 5, 4) and oe.shape == (3, 5, 4)
    assert deg[0].sum() == 0
    np.testing.assert_array_equal(oe[1, 0], [1, 2, 3, 4])
    np.testing.assert_array_equal(deg[1, :, 0], [0, 1, 1, 1, 1])
    np.testing.assert_array_equal(ie[2, 0], [1, 2, 3, 4])
    np.testing.assert_array_equal(ie[1, 1:, 0], [0, 0, 0, 0])


@pytest.mark.gpu
@pytest.mark.parametrize('tag', ['a', 'b', 'c'])
def test_sampler_init_tables_and_controls(tag):
    """the full DirectedCaseControlSampler.init path on the engine: tables exactly the
    reference's; control sets valid (distinct, non-neighbours, not self, -1 padding)"""
    from dynetlsm_amd import Chain, DirectedCaseControlSampler
    g = load_golden('likelihoods.npz')
    Y = g[tag + '_Yd']
    T, N, _ = Y.shape
    n_control = 3
    with Chain(T, N, 2, 'case_control', seed=11) as c:
        s = DirectedCaseControlSampler(n_control=n_control, n_resample=100, chain=c).init(Y)
        np.testing.assert_array_equal(s.degrees_, g[tag + '_degrees'])
        np.testing.assert_array_equal(s.in_edges_, g[tag + '_in_edges'])
        np.testing.assert_array_equal(s.out_edges_, g[tag + '_out_edges'])
        assert s.n_iter == 1
        ci, co = s.control_nodes_in_, s.control_nodes_out_
        assert ci.shape == (T, N, n_control) == co.shape
        for t in range(T):
            for i in range(N):
                for arr, nb in ((co, Y[t, i] == 1), (ci, Y[t, :, i] == 1)):
                    v = arr[t, i][arr[t, i] >= 0]
                    n_zero = N - 1 - int(nb.sum())
                    assert v.size == min(n_control, n_zero)
                    assert np.unique(v).size == v.size
                    assert i not in v and not nb[v].any()
                    assert (arr[t, i][v.size:] == -1).all()
        # the fraction form of n_control (case_control_likelihood.py:40-43)
    with Chain(T, N, 2, 'case_control', seed=11) as c:
        s = DirectedCaseControlSampler(n_control=0.5, n_resample=None, chain=c).init(Y)
        assert s.n_control_ == int(0.5 * N)
        s.resample(1)                         # n_resample None: never redraws
        assert s.n_iter == 2
