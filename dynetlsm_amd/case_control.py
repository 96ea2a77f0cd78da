"""DirectedCaseControlSampler (case_control_likelihood.py:36-112) on the engine.

``init`` builds the degree table and the zero-padded in / out edge lists on the
host (vectorised numpy; the reference's double Python loop is O(T N^2)) and
uploads them; control nodes are drawn on the device
(``dlsm_resample_controls``), so at N = 10^4 a resample takes microseconds
instead of minutes.
"""
import numbers

import numpy as np

__all__ = ['DirectedCaseControlSampler', 'build_edge_lists']


def build_edge_lists(Y):
    """degrees_[T, N, 2] (col 0 in-degree, col 1 out-degree) and the padded
    in_edges_ / out_edges_ of case_control_likelihood.py:45-68."""
    T, N, _ = Y.shape
    A = Y == 1
    deg = np.zeros((T, N, 2), dtype=np.int64)
    deg[:, :, 0] = A.sum(axis=1)
    deg[:, :, 1] = A.sum(axis=2)
    in_edges = np.zeros((T, N, int(deg[:, :, 0].max())), dtype=np.int64)
    out_edges = np.zeros((T, N, int(deg[:, :, 1].max())), dtype=np.int64)
    for t in range(T):
        r, c = np.nonzero(A[t])                 # row-major: sorted by r then c
        pos = np.arange(r.size) - np.repeat(np.cumsum(deg[t, :, 1]) - deg[t, :, 1],
                                            deg[t, :, 1])
        out_edges[t, r, pos] = c
        cT, rT = np.nonzero(A[t].T)             # sorted by target then source
        pos = np.arange(cT.size) - np.repeat(np.cumsum(deg[t, :, 0]) - deg[t, :, 0],
                                             deg[t, :, 0])
        in_edges[t, cT, pos] = rT
    return deg, in_edges, out_edges


class DirectedCaseControlSampler(object):
    def __init__(self, n_control=100, n_resample=100, chain=None):
        self.n_control = n_control
        self.n_resample = n_resample
        self.chain = chain
        self.n_iter = 0

    def init(self, Y):
        T, N, _ = Y.shape
        if isinstance(self.n_control, (numbers.Integral, np.integer)):
            self.n_control_ = int(self.n_control)
        else:
            self.n_control_ = int(self.n_control * N)
        self.degrees_, self.in_edges_, self.out_edges_ = build_edge_lists(Y)
        self.chain.upload_edges(self.in_edges_, self.out_edges_, self.degrees_)
        self.sample(0)
        self.n_iter += 1
        return self

    def sample(self, it):
        self.chain.resample_controls(it, self.n_control_)

    def resample(self, it):
        """case_control_likelihood.py:27-33 (cadence kept; draws keyed by ``it``)"""
        if self.n_resample is not None and self.n_iter % self.n_resample == 0:
            self.sample(it)
        self.n_iter += 1

    @property
    def control_nodes_in_(self):
        return self.chain.get_controls()[0]

    @property
    def control_nodes_out_(self):
        return self.chain.get_controls()[1]
