"""Edge cases of the HIP path against the CPU oracle: shortest sequences, the largest supported dimensions, ragged chain
counts (N not a multiple of the 16-chain workgroup tile), one hidden dimension, non positive definite K_mm."""
import numpy as np
import pytest
import torch

from cbfssm import synthetic as syn
from cbfssm.hip import ops, train, lib

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _run(kw, condition=True, grads=True, scale=0.1):
    from oracle import cbfssm_oracle as orc
    from oracle import cbfssm_torch_ref as tref
    w = syn.tiny(**kw)
    cfg = w.model_config()
    p = syn.perturb_params(syn.make_params(w, seed=1), scale=scale)
    u, y = syn.make_inputs(w)
    noise = syn.make_noise(w)
    ref = orc.elbo_step(cfg, p, u, y, noise, condition)
    eng = train.HipElboGrad(cfg, DEV)
    params = {k: torch.tensor(v, device=DEV) for k, v in p.items()}
    loss, terms, ws = eng.forward(params, u, y, noise, condition)
    assert float(terms['info']) == 0.0
    assert float(loss) == pytest.approx(ref['loss'], rel=1e-9)
    np.testing.assert_allclose(ops.as_btsd(ws.x, w.B, w.S).cpu().numpy(), ref['x_final'], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(ws.pred_mean.cpu().numpy(), ref['pred_mean'], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(ws.pred_var.cpu().numpy(), ref['pred_var'], rtol=1e-8, atol=1e-12)
    if grads:
        loss2, g, _ = eng.loss_and_grads(params, u, y, noise, condition)
        scal, gref = tref.loss_and_grads(cfg, p, u, y, noise, condition)
        assert float(loss2) == pytest.approx(scal['loss'], rel=1e-9)
        for k in train.PARAM_NAMES:
            err = np.abs(g[k].cpu().numpy() - gref[k]).max() / (np.abs(gref[k]).max() + 1e-300)
            assert err < 1e-6, (k, err)


@pytest.mark.parametrize('T', [1, 2, 3])
def test_shortest_sequences(T):
    _run(dict(T=T, B=2, S=5, M=12, recog_len=3))


def test_maximum_dimensions():
    # dim_x = 16 (one MFMA row block of outputs), D = dim_x + dim_u = 24 (six k-steps), one hidden dimension
    _run(dict(dim_x=16, dim_u=8, dim_y=15, T=6, B=1, S=7, M=33, recog_len=2))


def test_single_output_dim_and_m1():
    _run(dict(dim_x=2, dim_u=1, dim_y=1, T=5, B=2, S=3, M=1, recog_len=1))


@pytest.mark.parametrize('B,S', [(1, 1), (1, 17), (3, 11), (2, 16)])
def test_ragged_chain_counts(B, S):
    _run(dict(T=7, B=B, S=S, M=20, recog_len=2), grads=(B * S != 1))


def test_largest_tile_eval_and_condition_false():
    _run(dict(M=320, dim_x=4, dim_u=2, dim_y=2, T=6, B=1, S=5, recog_len=2), condition=False, grads=False)


def test_limits_are_errors_not_fallbacks():
    with pytest.raises(lib.CbfssmHipError):
        ops.GPPack(321, 5, 4, DEV)
    with pytest.raises(lib.CbfssmHipError):
        ops.GPPack(10, 25, 4, DEV)
    w = syn.tiny(dim_x=3, dim_y=3, dim_u=1)           # no hidden dimension: backward GP has zero outputs
    with pytest.raises(lib.CbfssmHipError):
        train.HipElboGrad(w.model_config(), DEV)


def test_non_pd_kmm_raises_through_the_model_surface(tmp_path):
    from cbfssm.model import CBFSSM, Session, InvalidArgumentError
    from cbfssm.datasets import make_synthetic_ds
    ds_cls = make_synthetic_ds(1, 1, 200, 80)
    cfg = syn.tiny(dim_u=1, dim_y=1, dim_x=3, M=8).model_config(ds_cls)
    cfg['seed'] = 1
    model = CBFSSM(cfg)
    ds = ds_cls(20, 10)
    with Session() as sess:
        sess.run(model.init)
        # a parameter that went NaN during training: the leading minor test `pivot > 0` fails, as tf.cholesky does
        model._opt.views['f.zeta_pos'][0, 0] = float('nan')
        model.load_ds(sess, ds.train_in_batch, ds.train_out_batch)
        with pytest.raises(InvalidArgumentError):
            sess.run(model.loss, {model.condition: True})
