"""Wall time of `Outputs.create_all` (reference cbfssm/outputs/outputs.py:36-164) on a Sarcos-shaped synthetic dataset with
several test experiments: the reference's loop of one B = 1 `sess.run` per experiment (CBFSSM_OUTPUTS_LOOP=1) against
the batched evaluation (`model.run_experiments`: all experiments in one launch, each with the noise its own run would
have drawn).  Same seed in both runs: mse.txt must agree to 1e-10.

    python profiles/tools/outputs_walltime.py [n_test_experiments] [test_len]
"""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'cbf-ssm_amd')]
import numpy as np
import torch
from cbfssm.datasets.base_ds import BaseDS
from cbfssm.training import Trainer
from cbfssm.outputs import Outputs
from cbfssm.model import CBFSSM
from cbfssm import synthetic as syn

n_exp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
t_len = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
w = syn.WORKLOADS['C3']


class SarcosShaped(BaseDS):
    """dim_u = dim_y = 7; one long training experiment, n_exp test experiments of t_len steps (seeded)"""
    dim_u, dim_y = w.dim_u, w.dim_y

    def __init__(self, seq_len, seq_stride):
        BaseDS.__init__(self, seq_len, seq_stride)
        rng = np.random.default_rng(1)
        n_train = 4000

        def series(n):
            u = np.cumsum(rng.standard_normal((n, self.dim_u)), axis=0) * 0.05
            h = np.zeros(self.dim_y)
            y = np.zeros((n, self.dim_y))
            A = 0.9 * np.linalg.qr(rng.standard_normal((self.dim_y, self.dim_y)))[0]
            Bm = 0.3 * rng.standard_normal((self.dim_y, self.dim_u))
            for i in range(n):
                h = np.tanh(A @ h + Bm @ u[i])
                y[i] = h + 0.02 * rng.standard_normal(self.dim_y)
            return u, y
        u, y = series(n_train)
        self.normalize_init(u, y)
        self.train_in, self.train_out = self.normalize(u, 'in')[None], self.normalize(y, 'out')[None]
        tests = [series(t_len) for _ in range(n_exp)]
        self.test_in = np.stack([self.normalize(a, 'in') for a, _ in tests])
        self.test_out = np.stack([self.normalize(b, 'out') for _, b in tests])
        self.create_batches()


cfg = dict(w.model_config())
cfg.update({'ds': SarcosShaped, 'batch_size': 32, 'shuffle': 10000, 'seed': 5})
ds = SarcosShaped(w.T, w.T)
res = {}
with tempfile.TemporaryDirectory() as d:
    model = CBFSSM(cfg)
    Trainer(model, d).train(ds, 1)
    for tag, env in (('loop (reference: one B=1 run per experiment)', '1'), ('batched (one launch)', '')):
        if env:
            os.environ['CBFSSM_OUTPUTS_LOOP'] = env
        else:
            os.environ.pop('CBFSSM_OUTPUTS_LOOP', None)
        times = []
        for rep in range(2):                                   # second repetition: allocations and first launches done
            m = CBFSSM(cfg)                                    # same seed -> same noise stream in both variants
            out = Outputs(os.path.join(d, 'out_' + ('loop' if env else 'batch')))
            out.set_ds(ds)
            out.set_model(m, d)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out.create_all()
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
        res[tag] = (times, open(os.path.join(out.out_dir, 'mse.txt')).read(), out.get_last_rmse())
print('\nSarcos-shaped synthetic dataset: %d test experiments x %d steps, M=%d S=%d dim_x=%d' % (n_exp, t_len, w.M, w.S, w.dim_x))
for tag, (times, txt, rmse) in res.items():
    print('%-48s create_all %.3f s (first call %.3f s)   RMSE %.12f' % (tag, times[-1], times[0], rmse))
a, b = [r[2] for r in res.values()]
print('RMSE difference %.3e (relative %.3e)' % (abs(a - b), abs(a - b) / abs(a)))
assert abs(a - b) <= 1e-10 * abs(a), 'batched evaluation changed the result'
