"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and the committed golden fixtures.

Tolerance: BASELINE.json's north_star asks for 1e-5 relative on the ELBO and the predictive mean/variance; the
float64 kernels are held to 1e-9 on scalars and 1e-8 on trajectories here (both paths are float64, they differ
only in summation order, in K^-1-contraction vs two triangular solves and in exp(a+b) vs exp(a)*exp(b))."""
import os
import numpy as np
import pytest
import torch

from cbfssm import synthetic as syn
from cbfssm.hip import ops, lib

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')
DEV = 'cuda:0'


def _oracle():
    from oracle import cbfssm_oracle as orc
    return orc


def _gp_args(p, g):
    orc = _oracle()
    return (p[g + '.zeta_pos'], orc.tf_forward(p[g + '.lengthscales_unc']), orc.tf_forward(p[g + '.variance_unc']),
            p[g + '.zeta_mean'], orc.tf_forward(p[g + '.zeta_var_unc']))


@pytest.mark.parametrize('M,D', [(5, 3), (20, 5), (33, 7), (100, 21), (200, 21), (300, 6)])
def test_kmm_chol(M, D):
    orc = _oracle()
    rng = np.random.default_rng(M)
    Z = rng.uniform(-2, 2, (M, D))
    ls = rng.uniform(0.8, 2.0, D)
    var = np.array([0.3])
    Kmm, L, info = ops.kmm_chol(torch.tensor(Z, device=DEV), torch.tensor(ls, device=DEV), torch.tensor(var, device=DEV))
    kern = orc.RBF(orc.tf_backward(var), orc.tf_backward(ls))
    K_ref = kern.K(Z)
    L_ref = orc.cast_cholesky(K_ref)
    assert float(info[0]) == 0.0
    np.testing.assert_allclose(Kmm.cpu().numpy(), K_ref, rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(L.cpu().numpy(), L_ref, rtol=1e-7, atol=1e-10)
    Lg = L.cpu().numpy()
    np.testing.assert_allclose(Lg @ Lg.T, K_ref + 1e-8 * np.eye(M), rtol=1e-11, atol=1e-13)
    assert np.all(np.triu(Lg, 1) == 0.0)


def test_kmm_chol_reports_non_pd():
    # duplicated inducing points with zero jitter: K is singular, the factorisation must flag a leading minor
    Z = np.tile(np.random.default_rng(0).standard_normal((1, 3)), (6, 1))
    _, _, info = ops.kmm_chol(torch.tensor(Z, device=DEV), torch.ones(3, device=DEV, dtype=torch.float64),
                              torch.ones(1, device=DEV, dtype=torch.float64), jitter=-1e-3)
    assert float(info[0]) >= 1.0


@pytest.mark.parametrize('M,dim_x,dim_u,dim_y', [(12, 5, 2, 2), (20, 4, 1, 1), (50, 4, 1, 1), (100, 14, 7, 7),
                                                (130, 9, 3, 2), (200, 14, 7, 7), (250, 4, 2, 2), (300, 4, 2, 2)])
def test_gp_prepare_and_predict(M, dim_x, dim_u, dim_y):
    orc = _oracle()
    w = syn.tiny(M=M, dim_x=dim_x, dim_u=dim_u, dim_y=dim_y)
    p = syn.perturb_params(syn.make_params(w, seed=M))
    rng = np.random.default_rng(7)
    for g, Do in (('f', dim_x), ('b', dim_x - dim_y)):
        args = _gp_args(p, g)
        gp = orc.GPModel(p[g + '.zeta_pos'], p[g + '.zeta_mean'], p[g + '.zeta_var_unc'], p[g + '.variance_unc'],
                         p[g + '.lengthscales_unc'])
        pack = ops.GPPack(M, w.D, Do, DEV).prepare(*[torch.tensor(a, device=DEV) for a in args])
        scal = pack.scal.cpu().numpy()
        assert scal[lib.SCAL_INFO] == 0.0
        K = gp.kern.K(gp.zeta_pos) + 1e-8 * np.eye(M)
        np.testing.assert_allclose(pack.Kinv.cpu().numpy() @ K, np.eye(M), atol=1e-6)
        assert scal[lib.SCAL_LOGDET] == pytest.approx(2 * np.sum(np.log(np.diag(gp.cholesky))), rel=1e-10)
        assert scal[lib.SCAL_KLZ] == pytest.approx(gp.prior_kl(), rel=1e-9)
        for npts in (1, 16, 37):
            X = rng.standard_normal((npts, w.D)) * 1.5
            fm, fv = pack.predict(torch.tensor(X, device=DEV))
            fm_ref, fv_ref = gp.predict(X)
            np.testing.assert_allclose(fm.cpu().numpy(), fm_ref, rtol=1e-8, atol=1e-11)
            np.testing.assert_allclose(fv.cpu().numpy(), fv_ref, rtol=1e-8, atol=1e-11)


def _load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    wk = {k[len('workload_'):]: z[k] for k in z.files if k.startswith('workload_')}
    kw = {k: (tuple(v.tolist()) if v.ndim else v.item()) for k, v in wk.items()}
    w = syn.Workload(name, **kw)
    p = {k[len('param_'):]: z[k] for k in z.files if k.startswith('param_')}
    noise = {k[len('noise_'):]: z[k] for k in z.files if k.startswith('noise_')}
    return z, w, p, noise


def _compare(ws, w, ref, rtol_s=1e-9, rtol_t=1e-8):
    out = ws.out.cpu().numpy()
    assert out[7] == 0.0
    for i, k in enumerate(('loglik', 'kl_x', 'entropy', 'kl_z_f', 'kl_z_b')):
        assert out[i] == pytest.approx(float(ref[k]), rel=rtol_s, abs=1e-9), k
    assert out[6] == pytest.approx(float(ref['loss']), rel=rtol_s)
    x_final = ops.as_btsd(ws.x, w.B, w.S).cpu().numpy()
    np.testing.assert_allclose(x_final, ref['x_final'], rtol=rtol_t, atol=1e-10)
    y2 = ops.as_btsd(ws.y2, w.B, w.S).cpu().numpy()
    np.testing.assert_allclose(y2, ref['y_tilde'][..., w.dim_y:], rtol=rtol_t, atol=1e-10)
    np.testing.assert_allclose(ws.pred_mean.cpu().numpy(), ref['pred_mean'], rtol=rtol_t, atol=1e-10)
    np.testing.assert_allclose(ws.pred_var.cpu().numpy(), ref['pred_var'], rtol=rtol_t, atol=1e-12)


@pytest.mark.parametrize('name', ['tiny', 'mini_sarcos', 'mini_smallscale'])
@pytest.mark.parametrize('cond', [True, False])
def test_elbo_forward_matches_golden(name, cond):
    z, w, p, noise = _load_golden(name)
    eng = ops.HipElbo(w.model_config(), DEV)
    eng.prepare(p)
    ws = eng.run(z['u'], z['y'], noise, condition=cond)
    tag = 'c1_' if cond else 'c0_'
    ref = {k[3:]: z[k] for k in z.files if k.startswith(tag)}
    _compare(ws, w, ref)


@pytest.mark.parametrize('kw', [
    dict(M=100, dim_x=14, dim_u=7, dim_y=7, T=20, B=2, S=20, recog_len=4, k_factor=50., var_y=0.05 ** 2),   # Sarcos tile
    dict(M=200, dim_x=14, dim_u=7, dim_y=7, T=9, B=1, S=20, recog_len=2, k_factor=50.),                     # C4 tile
    dict(M=300, dim_x=4, dim_u=2, dim_y=2, T=9, B=2, S=9, recog_len=50, k_factor=1.),                       # C5 tile, T < R
    dict(M=20, dim_x=4, dim_u=1, dim_y=1, T=50, B=3, S=50, recog_len=16, k_factor=100., gp_len=2.),         # C1 tile
    dict(M=50, dim_x=4, dim_u=1, dim_y=1, T=33, B=1, S=1, recog_len=16, k_factor=100.),                     # single chain
])
def test_elbo_forward_matches_oracle(kw):
    orc = _oracle()
    w = syn.tiny(**kw)
    cfg = w.model_config()
    p = syn.perturb_params(syn.make_params(w, seed=1), scale=0.1)
    u, y = syn.make_inputs(w, seed=0)
    noise = syn.make_noise(w, seed=2)
    ref = orc.elbo_step(cfg, p, u, y, noise, True)
    eng = ops.HipElbo(cfg, DEV)
    eng.prepare(p)
    ws = eng.run(u, y, noise, condition=True)
    _compare(ws, w, ref)
    # determinism: a second evaluation is bit-identical (fixed-order partial sums, no atomics)
    out1 = ws.out.clone()
    x1 = ws.x.clone()
    ws = eng.run(u, y, noise, condition=True)
    assert torch.equal(out1, ws.out) and torch.equal(x1, ws.x)


@pytest.mark.parametrize('kw', [
    dict(M=200, dim_x=14, dim_u=7, dim_y=7, T=9, B=1, S=20, recog_len=2, k_factor=50.),      # 2 column blocks, 2nd ragged
    dict(M=250, dim_x=4, dim_u=2, dim_y=2, T=9, B=3, S=11, recog_len=3, k_factor=1.),        # 3 blocks: odd count
])
def test_shared_operand_pass_variant_matches_oracle(kw, monkeypatch):
    """Tile heights 13..16: pass_kernel<NC = 2> (two column blocks share every streamed K^-1 operand load; the default
    from N = 4096 chains on) forced on small cases, against the oracle, and the adjoint fed by its saved A2 tiles."""
    from cbfssm.hip import train
    from oracle import cbfssm_torch_ref as tref
    monkeypatch.setenv('CBFSSM_NC_FWD', '3')
    monkeypatch.setenv('CBFSSM_NC_BWD', '3')
    orc = _oracle()
    w = syn.tiny(loss_factors=(2., 0.4), **kw)
    cfg = w.model_config()
    p = syn.perturb_params(syn.make_params(w, seed=1), scale=0.1)
    u, y = syn.make_inputs(w, seed=0)
    noise = syn.make_noise(w, seed=2)
    ref = orc.elbo_step(cfg, p, u, y, noise, True)
    eng = ops.HipElbo(cfg, DEV)
    eng.prepare(p)
    ws = eng.run(u, y, noise, condition=True)
    _compare(ws, w, ref)
    g = train.HipElboGrad(cfg, DEV)
    loss, grads, _ = g.loss_and_grads({k: torch.tensor(v, device=DEV) for k, v in p.items()}, u, y, noise)
    scal, gref = tref.loss_and_grads(cfg, p, u, y, noise, True)
    assert abs(float(loss) - scal['loss']) <= 1e-9 * abs(scal['loss'])
    for k in train.PARAM_NAMES:
        np.testing.assert_allclose(grads[k].cpu().numpy(), gref[k], rtol=1e-6, atol=1e-7 * np.abs(gref[k]).max())


@pytest.mark.parametrize('kw', [
    dict(M=100, dim_x=14, dim_u=7, dim_y=7, T=20, B=3, S=20, recog_len=4, k_factor=50., var_y=0.05 ** 2),   # Sarcos tile, 4 blocks
    dict(M=20, dim_x=4, dim_u=1, dim_y=1, T=23, B=3, S=11, recog_len=5, k_factor=100.),                     # 3 blocks: odd count
])
def test_skewed_pass_variant_matches_oracle(kw, monkeypatch):
    """pass_kernel_skew (two column blocks half a step apart; opt-in since round 2) forced on, against the oracle."""
    monkeypatch.setenv('CBFSSM_NC_FWD', '2')
    monkeypatch.setenv('CBFSSM_NC_BWD', '2')
    monkeypatch.setenv('CBFSSM_GP_FORM', 'dense')
    orc = _oracle()
    w = syn.tiny(loss_factors=(2., 0.4), **kw)
    cfg = w.model_config()
    p = syn.perturb_params(syn.make_params(w, seed=1), scale=0.1)
    u, y = syn.make_inputs(w, seed=0)
    noise = syn.make_noise(w, seed=2)
    ref = orc.elbo_step(cfg, p, u, y, noise, True)
    eng = ops.HipElbo(cfg, DEV)
    eng.prepare(p)
    ws = eng.run(u, y, noise, condition=True)
    _compare(ws, w, ref)
