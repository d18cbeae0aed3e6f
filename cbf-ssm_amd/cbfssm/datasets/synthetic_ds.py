"""A seeded synthetic dataset with the BaseDS interface (no file needed): a stable nonlinear state-space system,
for tests, smoke runs and benchmarks on machines without the reference's data files."""
import numpy as np

from .base_ds import BaseDS


def make_synthetic_ds(dim_u=1, dim_y=1, n_train=1200, n_test=400, seed=0, name='SyntheticDS'):
    def __init__(self, seq_len, seq_stride):
        BaseDS.__init__(self, seq_len, seq_stride)
        rng = np.random.default_rng(seed)
        n = n_train + n_test
        dim_h = max(dim_y, 2)
        A = 0.9 * np.linalg.qr(rng.standard_normal((dim_h, dim_h)))[0]
        Bm = 0.5 * rng.standard_normal((dim_h, dim_u))
        C = rng.standard_normal((dim_y, dim_h))
        u = np.cumsum(rng.standard_normal((n, dim_u)), axis=0) * 0.1
        u = u - u.mean(0)
        h = np.zeros(dim_h)
        y = np.zeros((n, dim_y))
        for i in range(n):
            y[i] = C @ h + 0.05 * rng.standard_normal(dim_y)
            h = np.tanh(A @ h + Bm @ u[i])
        self.normalize_init(u[:n_train], y[:n_train])
        u, y = self.normalize(u, 'in'), self.normalize(y, 'out')
        self.train_in, self.train_out = u[None, :n_train], y[None, :n_train]
        self.test_in, self.test_out = u[None, n_train:], y[None, n_train:]
        self.create_batches()
    return type(name, (BaseDS,), {'dim_u': dim_u, 'dim_y': dim_y, '__init__': __init__})
