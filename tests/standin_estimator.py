"""A stand-in for an estimator, for the CPU tests of multichain.fit_chains: same surface
(random_state / chain_id / device attributes, fit(Y, init), fitted traces), no GPU."""
import numpy as np


class StandInEstimator(object):
    def __init__(self, n_iter=60, random_state=0, chain_id=0, device=0, fail_on_chain=None, thin=None,
                 burn=10, hang_on_chain=None):
        self.n_iter, self.random_state, self.chain_id, self.device = n_iter, random_state, chain_id, device
        self.fail_on_chain, self.thin, self.burn, self.hang_on_chain = fail_on_chain, thin, burn, hang_on_chain

    @property
    def n_burn_(self):
        return self.burn        # in ITERATIONS, as the estimators' n_burn_ (burn + tune)

    def fit(self, Y, init=None):
        if self.fail_on_chain is not None and self.chain_id == self.fail_on_chain:
            raise RuntimeError('stand-in failure on chain %d' % self.chain_id)
        if self.hang_on_chain is not None and self.chain_id == self.hang_on_chain:
            import time
            time.sleep(3600)
        T, N = Y.shape[:2]
        rng = np.random.RandomState(1000 * int(self.random_state) + int(self.chain_id))
        e = rng.randn(self.n_iter)
        lp = np.empty(self.n_iter)
        lp[0] = e[0]
        for i in range(1, self.n_iter):
            lp[i] = 0.5 * lp[i - 1] + e[i]
        self.logps_ = lp + Y.sum() + (0.0 if init is None else float(init['shift']))
        self.intercepts_ = rng.randn(self.n_iter, 1)
        if self.thin:           # the estimators store every thin-th row (hdp_lpcm.py:1072-1083)
            self.logps_, self.intercepts_ = self.logps_[::self.thin], self.intercepts_[::self.thin]
        self.X_ = rng.randn(T, N, 2)
        self.z_ = rng.randint(0, 3, size=(T, N))
        self.seen_missing_ = int((Y == -1).sum())
        return self
