"""CBFSSMHALF on the GPU against the CPU oracle (cbfssm/model/cbfssmhalf.py restated in oracle/): loss terms,
trajectories, predictive moments and the gradient of every trainable tensor incl. the GRU recognition model."""
import os
import numpy as np
import pytest
import torch

from cbfssm import synthetic as syn
from cbfssm.hip import ops
from cbfssm.hip.train_half import HipHalfGrad, half_param_names

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _setup(recog, **kw):
    from test_oracle import _half_setup
    return _half_setup(recog, **kw)


@pytest.mark.parametrize('recog,kw', [
    ('rnn', dict(T=11, B=3, S=4, M=12, recog_len=3)),
    ('output', dict(T=9, B=2, S=7, M=20, recog_len=4, k_factor=20.)),
    ('rnn', dict(T=1, B=2, S=5, M=12, recog_len=3)),
    ('rnn', dict(T=8, B=1, S=20, M=100, dim_x=14, dim_u=7, dim_y=7, recog_len=2, k_factor=50.)),     # Sarcos tile
    ('rnn', dict(T=7, B=2, S=6, M=12, dim_x=3, dim_u=2, dim_y=3, recog_len=2)),                     # no hidden dims
    ('rnn', dict(T=9, B=2, S=9, M=130, dim_x=6, dim_u=2, dim_y=2, recog_len=2, adjoint=3e-4)),      # stash mode, chunked
])
@pytest.mark.parametrize('cond', [True, False])
def test_half_matches_oracle(recog, kw, cond):
    from oracle import cbfssm_oracle as orc
    from oracle import cbfssm_torch_ref as tref
    kw = dict(kw)
    gib = kw.pop('adjoint', None)
    w, cfg, p, u, y, noise = _setup(recog, **kw)
    if gib:
        cfg['adjoint_stash_gib'] = gib
    ref = orc.CBFSSMHALFOracle(cfg, p).run(u, y, noise, cond)
    eng = HipHalfGrad(cfg, DEV)
    params = {k: torch.tensor(v, device=DEV) for k, v in p.items()}
    loss, terms, ws = eng.forward(params, u, y, noise, cond)
    assert float(terms['info']) == 0.0
    assert float(loss) == pytest.approx(ref['loss'], rel=1e-9)
    np.testing.assert_allclose(ops.as_btsd(ws.x, w.B, w.S).cpu().numpy(), ref['x_final'], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(ws.pred_mean.cpu().numpy(), ref['pred_mean'], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(ws.pred_var.cpu().numpy(), ref['pred_var'], rtol=1e-8, atol=1e-12)
    loss2, grads, _ = eng.loss_and_grads(params, u, y, noise, cond)
    lref, gref = tref.half_loss_and_grads(cfg, p, u, y, noise, cond)
    assert float(loss2) == pytest.approx(lref, rel=1e-9)
    assert set(grads) == set(half_param_names(cfg))
    for k in grads:
        err = np.abs(grads[k].cpu().numpy() - gref[k]).max() / (np.abs(gref[k]).max() + 1e-300)
        assert err < 1e-6, (k, err)


def test_half_template_flow(tmp_path):
    """run/template.py:17 says: use CBFSSMHALF when there is no unstable hidden dimension."""
    from cbfssm.datasets import make_synthetic_ds
    from cbfssm.training import Trainer
    from cbfssm.outputs import Outputs
    from cbfssm.model import CBFSSMHALF
    root_dir = str(tmp_path / 'half')
    ds_sel = make_synthetic_ds(dim_u=1, dim_y=1, n_train=400, n_test=160, seed=2)
    dim_x = 3
    cfg = {'ds': ds_sel, 'batch_size': 8, 'shuffle': 10000, 'seed': 7, 'dim_x': dim_x, 'ind_pnt_num': 20,
           'samples': 10, 'learning_rate': 0.05, 'loss_factors': np.asarray([1., 0.]), 'k_factor': 5.,
           'recog_len': 8, 'zeta_pos': 2., 'zeta_mean': 0.05 ** 2, 'zeta_var': 0.01 ** 2,
           'var_x': np.asarray([0.002 ** 2] * dim_x), 'var_y': np.asarray([1. ** 2] * ds_sel.dim_y),
           'gp_var': 0.5 ** 2, 'gp_len': 2.}
    outputs = Outputs(root_dir)
    ds = ds_sel(40, 20)
    outputs.set_ds(ds)
    model = CBFSSMHALF(cfg)
    outputs.set_model(model, root_dir)
    trainer = Trainer(model, root_dir)
    trainer.train(ds, 4)
    outputs.set_trainer(trainer)
    outputs.create_all()
    assert trainer.train_all[-1] < trainer.train_all[0] and all(np.isfinite(trainer.test_all))
    assert np.isfinite(outputs.get_last_rmse())
    assert os.path.isfile(os.path.join(root_dir, 'var_dump.txt'))


@pytest.mark.parametrize('cond', [True, False])
def test_half_matches_golden(cond):
    z = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'half_tiny.npz'))
    wk = {k[len('workload_'):]: z[k] for k in z.files if k.startswith('workload_')}
    w = syn.Workload('half_tiny', **{k: (tuple(v.tolist()) if v.ndim else v.item()) for k, v in wk.items()})
    cfg = w.model_config()
    cfg['var_y'] = z['var_y_cfg']
    cfg['recog_model'] = 'rnn'
    p = {k[len('param_'):]: z[k] for k in z.files if k.startswith('param_')}
    eng = HipHalfGrad(cfg, DEV)
    params = {k: torch.tensor(v, device=DEV) for k, v in p.items()}
    tag = 'c1_' if cond else 'c0_'
    loss, grads, terms = eng.loss_and_grads(params, z['u'], z['y'], {'eps_f': z['noise_eps_f']}, cond)
    assert float(loss) == pytest.approx(float(z[tag + 'loss']), rel=1e-9)
    assert float(terms['kl_x']) == pytest.approx(float(z[tag + 'kl_x']), rel=1e-9, abs=1e-9)
    np.testing.assert_allclose(ops.as_btsd(eng.last_ws.x, w.B, w.S).cpu().numpy(), z[tag + 'x_final'], rtol=1e-8, atol=1e-10)
    for k in grads:
        r = z[tag + 'grad_' + k]
        assert np.abs(grads[k].cpu().numpy() - r).max() / (np.abs(r).max() + 1e-300) < 1e-6, k
