"""PR-SSM baseline (reference cbfssm/model/prssm.py) on the GPU against its CPU restatement."""
import os
import numpy as np
import pytest
import torch

from cbfssm.hip import ops
from cbfssm.hip.train_half import HipHalfGrad, half_param_names

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.mark.parametrize('recog,kw', [
    ('output', dict(T=9, B=2, S=7, M=20, recog_len=4)),
    ('rnn', dict(T=11, B=3, S=4, M=12, recog_len=3)),
    ('conv', dict(T=20, B=2, S=5, M=33, recog_len=16, dim_x=4, dim_u=1, dim_y=1)),
    ('rnn', dict(T=8, B=2, S=9, M=130, dim_x=6, dim_u=2, dim_y=2, recog_len=2)),      # stash mode
])
def test_prssm_matches_restatement(recog, kw):
    from test_oracle import _prssm_setup
    from oracle import cbfssm_torch_ref as tref
    w, cfg, p, u, y, noise = _prssm_setup(recog, **kw)
    ref, gref = tref.prssm_loss_and_grads(cfg, p, u, y, noise)
    eng = HipHalfGrad(cfg, DEV, variant='prssm')
    params = {k: torch.tensor(v, device=DEV) for k, v in p.items()}
    loss, terms, ws = eng.forward(params, u, y, noise, True)
    assert float(terms['info']) == 0.0
    # the conv recogniser runs in float32 on both sides (prssm.py:146,153): x_0 agrees to float32 rounding only
    assert float(loss) == pytest.approx(float(ref['loss']), rel=1e-9 if recog != 'conv' else 1e-6)
    tol = dict(rtol=1e-8, atol=1e-10) if recog != 'conv' else dict(rtol=1e-5, atol=5e-6)    # conv recogniser is float32
    np.testing.assert_allclose(ops.as_btsd(ws.x, w.B, w.S).cpu().numpy(), ref['x_final'], **tol)
    np.testing.assert_allclose(ws.pred_mean.cpu().numpy(), ref['pred_mean'], **tol)
    np.testing.assert_allclose(ws.pred_var.cpu().numpy(), ref['pred_var'], rtol=tol['rtol'], atol=tol['atol'])
    loss2, grads, _ = eng.loss_and_grads(params, u, y, noise, True)
    assert set(grads) == set(half_param_names(cfg, 'prssm'))
    for k in grads:
        err = np.abs(grads[k].cpu().numpy() - gref[k]).max() / (np.abs(gref[k]).max() + 1e-300)
        assert err < (1e-6 if recog != 'conv' else 1e-3), (k, err)


def test_prssm_template_flow(tmp_path):
    from cbfssm.datasets import make_synthetic_ds
    from cbfssm.training import Trainer
    from cbfssm.outputs import Outputs
    from cbfssm.model import PRSSM
    root_dir = str(tmp_path / 'prssm')
    ds_sel = make_synthetic_ds(dim_u=1, dim_y=1, n_train=400, n_test=160, seed=3)
    dim_x = 3
    cfg = {'ds': ds_sel, 'batch_size': 8, 'shuffle': 10000, 'seed': 9, 'dim_x': dim_x, 'ind_pnt_num': 20,
           'samples': 10, 'learning_rate': 0.05, 'loss_factors': np.asarray([0.1, 0.]), 'k_factor': 1.,
           'recog_len': 16, 'recog_model': 'conv', 'zeta_pos': 2., 'zeta_mean': 0.05 ** 2, 'zeta_var': 0.01 ** 2,
           'var_x': np.asarray([0.002 ** 2] * dim_x), 'var_y': np.asarray([1. ** 2] * ds_sel.dim_y),
           'gp_var': 0.5 ** 2, 'gp_len': 2.}
    outputs = Outputs(root_dir)
    ds = ds_sel(40, 20)
    outputs.set_ds(ds)
    model = PRSSM(cfg)
    outputs.set_model(model, root_dir)
    trainer = Trainer(model, root_dir)
    trainer.train(ds, 4)
    outputs.set_trainer(trainer)
    outputs.create_all()
    assert trainer.train_all[-1] < trainer.train_all[0] and all(np.isfinite(trainer.test_all))
    assert np.isfinite(outputs.get_last_rmse())
    assert 'kernel lengthscales:' in open(os.path.join(root_dir, 'var_dump.txt')).read()
