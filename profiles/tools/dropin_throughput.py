"""Throughput of the drop-in surface itself (reference run/template.py:50-64 flow, Trainer.train) on a C3-shaped
(or, with an argument, C1-/C2-shaped) synthetic dataset: what a user of `cbfssm.model.CBFSSM` + `cbfssm.training.Trainer` gets per train step, next to the
`bench.py` number for the bare engine."""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'cbf-ssm_amd')]
import numpy as np
import torch
from cbfssm.datasets import make_synthetic_ds
from cbfssm.training import Trainer
from cbfssm.model import CBFSSM

from cbfssm import synthetic as syn
w = syn.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else 'C3']           # C3 by default; C1 / C2: the launch-bound shapes
B, T, nb = w.B, w.T, 12
ds_sel = make_synthetic_ds(dim_u=w.dim_u, dim_y=w.dim_y, n_train=T * B * nb, n_test=T * B, seed=1)
cfg = dict(w.model_config())
cfg.update({'ds': ds_sel, 'batch_size': B, 'shuffle': 10000, 'seed': 5})
ds = ds_sel(T, T)
model = CBFSSM(cfg)
with tempfile.TemporaryDirectory() as d:
    tr = Trainer(model, d)
    tr.train(ds, 1)                        # warm-up epoch (allocations, first launches)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.train(ds, 2, retrain=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
n_train = (ds.train_in_batch.shape[0] + B - 1) // B
n_test = (ds.test_in_batch.shape[0] + B - 1) // B
print('epochs: 2 x (%d train batches + %d test batches of %d sequences) in %.3f s' % (n_train, n_test, B, dt))
print('=> %.2f ms per batch counting test batches as train batches (a train step costs ~3.6x an eval step)'
      % (dt / 2 / (n_train + n_test) * 1e3))
print('=> %.2f ms per train batch if the test pass cost nothing (upper bound on the train step)' % (dt / 2 / n_train * 1e3))
