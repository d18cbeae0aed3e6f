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


def test_train_tail_entry_points_through_the_abi():
    """cbfssm_constrain_f64, cbfssm_data_tail_f64 and cbfssm_adam_step_f64 called directly (ctypes) against numpy."""
    import ctypes as C
    from cbfssm.hip import lib, ops
    l = lib.load()
    rng = np.random.default_rng(3)
    M, dx, du, dy = 9, 5, 2, 2
    pl = lib.param_layout(M, dx, du, dy)
    n = int(pl.total)
    p = rng.standard_normal(n) * 3.0
    p[7] = 40.0          # softplus of a large argument
    p[8] = -40.0
    dp = torch.tensor(p, device=DEV)
    dc = torch.zeros_like(dp)
    st = ops._stream()
    lib.check(l.cbfssm_constrain_f64(C.byref(pl), ops._ptr(dp), ops._ptr(dc), st), 'constrain')
    unc = np.zeros(n, dtype=bool)
    for k in (2, 3, 4, 7, 8, 9, 10, 11):                      # the *_unc tensors
        hi = pl.off[k + 1] if k < 11 else n
        unc[pl.off[k]:hi] = True
    ref = np.where(unc, np.logaddexp(0.0, p) + 1e-10, p)      # tf_transform.py:19-21
    np.testing.assert_allclose(dc.cpu().numpy(), ref, rtol=1e-14, atol=1e-300)

    # Adam, TF 1.8 rule, three steps with fresh gradients
    m = torch.zeros_like(dp); v = torch.zeros_like(dp); t = torch.zeros(1, dtype=torch.float64, device=DEV)
    pn, mn, vn = p.copy(), np.zeros(n), np.zeros(n)
    lr, b1, b2, eps = 0.05, 0.9, 0.999, 1e-8
    for k in range(1, 4):
        g = rng.standard_normal(n)
        dg = torch.tensor(g, device=DEV)
        lib.check(l.cbfssm_adam_step_f64(n, ops._ptr(dp), ops._ptr(dg), ops._ptr(m), ops._ptr(v), ops._ptr(t), lr, b1, b2,
                                         eps, st), 'adam')
        mn = b1 * mn + (1 - b1) * g
        vn = b2 * vn + (1 - b2) * g * g
        pn = pn - lr * np.sqrt(1 - b2 ** k) / (1 - b1 ** k) * mn / (np.sqrt(vn) + eps)
    assert float(t[0]) == 3.0
    np.testing.assert_allclose(dp.cpu().numpy(), pn, rtol=1e-13, atol=1e-15)

    # data tail: per-dimension totals of the block partials -> d loss / d var_y   (cbfssm.py:245-251)
    B, T, S = 3, 7, 4
    prob = lib.make_problem(B, S, T, dx, du, dy, M, 2, 1.0, True)
    nll = int(l.cbfssm_loglik_partials(C.byref(prob)))
    llp = rng.standard_normal(nll)
    vy = rng.uniform(0.1, 2.0, dx)
    out8 = rng.standard_normal(8)
    tail = torch.zeros(3 + dy, dtype=torch.float64, device=DEV)
    cL = 0.37
    d_vy, d_ll, d_o8 = torch.tensor(vy, device=DEV), torch.tensor(llp, device=DEV), torch.tensor(out8, device=DEV)
    lib.check(l.cbfssm_data_tail_f64(C.byref(prob), ops._ptr(d_vy), ops._ptr(d_ll), ops._ptr(d_o8), cL, ops._ptr(tail), st),
              'data tail')
    ll_d = llp.reshape(-1, dy).sum(0)
    bts = B * T * S
    sq = (-2.0 * ll_d - bts * (np.log(2 * np.pi) + np.log(vy[:dy]))) * vy[:dy]
    ref_t = np.concatenate([out8[:3], -cL * 0.5 * (sq / vy[:dy] ** 2 - bts / vy[:dy])])
    np.testing.assert_allclose(tail.cpu().numpy(), ref_t, rtol=1e-12, atol=1e-14)


@pytest.mark.parametrize('S', [1, 2, 7, 50])
def test_loglik_moments_one_pass_keeps_the_digits(S):
    """cbfssm_loglik_moments_f64 takes the population moments over the particles (tf.nn.moments, cbfssm.py:267) in one
    pass over HBM (sums of differences to the first particle).  Worst case for a one-pass variance: particles that
    sit 1e6 spreads away from zero.  Checked against a long-double two-pass evaluation, called through the C ABI."""
    import ctypes as C
    l = lib.load()
    B, T, dx, dy = 3, 5, 6, 2
    rng = np.random.default_rng(S)
    x = 1e3 + 1e-3 * rng.standard_normal((T, B, S, dx))                  # (T, N, dim_x) as the forward pass writes it
    y = 1e3 + rng.standard_normal((B, T, dy))
    vy = np.abs(rng.standard_normal(dx)) + 0.1
    prob = lib.make_problem(B, S, T, dx, 1, dy, 8, 2, 1.0, True)
    f = dict(dtype=torch.float64, device=DEV)
    tx, ty, tv = torch.tensor(x, **f).contiguous(), torch.tensor(y, **f).contiguous(), torch.tensor(vy, **f)
    ll = torch.zeros(int(l.cbfssm_loglik_partials(C.byref(prob))), **f)
    pm, pv = torch.zeros(B, T, dy, **f), torch.zeros(B, T, dy, **f)
    im, iv = torch.zeros(B, T, dx, **f), torch.zeros(B, T, dx, **f)
    lib.check(l.cbfssm_loglik_moments_f64(C.byref(prob), ops._ptr(tv), ops._ptr(ty), ops._ptr(tx), ops._ptr(ll),
                                          ops._ptr(pm), ops._ptr(pv), ops._ptr(im), ops._ptr(iv), None), 'loglik')
    torch.cuda.synchronize()
    xl = np.transpose(x, (1, 0, 2, 3)).astype(np.longdouble)             # (B, T, S, dx)
    mean = xl.mean(axis=2)
    var = ((xl - mean[:, :, None, :]) ** 2).mean(axis=2)
    np.testing.assert_allclose(im.cpu().numpy(), mean.astype(np.float64), rtol=1e-15)
    np.testing.assert_allclose(iv.cpu().numpy(), var.astype(np.float64), rtol=1e-9, atol=1e-30)
    np.testing.assert_allclose(pv.cpu().numpy(), (var[..., :dy] + vy[:dy]).astype(np.float64), rtol=1e-12)
    lg = (-0.5 * (((y.astype(np.longdouble)[:, :, None, :] - xl[..., :dy]) ** 2) / vy[:dy]
                  + np.log(2 * np.pi * vy[:dy].astype(np.longdouble)))).sum()
    assert float(ll.sum()) == pytest.approx(float(lg), rel=1e-12)
