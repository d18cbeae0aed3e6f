"""The pass / predict kernels in the reference's own order of operations (gp_tf.py:137-145: A = L^-1 k,
fvar_0 = sigma^2 - |A|^2, A2 = L^-T A as two triangular products; layout.gp_form = CBFSSM_GP_FORM_TRI) against the
oracle and the committed fixtures, forced on for every tile height -- registers (M <= 112), streamed one row block per
wave (M <= 160), streamed two row blocks per wave (M <= 320) -- and the automatic choice between the two forms."""
import dataclasses
import os
import numpy as np
import pytest
import torch

from cbfssm import synthetic as syn
from cbfssm.hip import ops, lib, train

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.fixture
def tri(monkeypatch):
    monkeypatch.setenv('CBFSSM_GP_FORM', 'tri')


def _oracle():
    from oracle import cbfssm_oracle as orc
    return orc


@pytest.mark.parametrize('M,dim_x,dim_u,dim_y', [(5, 3, 1, 1), (12, 5, 2, 2), (20, 4, 1, 1), (50, 4, 1, 1), (100, 14, 7, 7),
                                                (112, 4, 2, 2), (130, 9, 3, 2), (160, 4, 1, 1), (200, 14, 7, 7),
                                                (250, 4, 2, 2), (300, 4, 2, 2), (320, 5, 1, 2)])
def test_gp_predict_tri_form(tri, M, dim_x, dim_u, dim_y):
    from test_hip_parity import _gp_args
    orc = _oracle()
    w = syn.tiny(M=M, dim_x=dim_x, dim_u=dim_u, dim_y=dim_y)
    p = syn.perturb_params(syn.make_params(w, seed=M))
    rng = np.random.default_rng(7)
    for g, Do in (('f', dim_x), ('b', dim_x - dim_y)):
        gp = orc.GPModel(p[g + '.zeta_pos'], p[g + '.zeta_mean'], p[g + '.zeta_var_unc'], p[g + '.variance_unc'],
                         p[g + '.lengthscales_unc'])
        pack = ops.GPPack(M, w.D, Do, DEV).prepare(*[torch.tensor(a, device=DEV) for a in _gp_args(p, g)])
        assert pack.gp_form() == 'tri'
        # the triangular operand images: W = L^-1 (lower), W^T
        Linv = np.linalg.inv(gp.cholesky)
        lay = pack.layout
        for name, ref in (('Wp', Linv), ('WTp', Linv.T)):
            img = pack.section(name, (lay.NBLK, lay.KS, 4, 16)).cpu().numpy()       # [rb][k-step][k in step][row]
            dense = img.transpose(0, 3, 1, 2).reshape(lay.Mp, lay.Mp)
            np.testing.assert_allclose(dense[:M, :M], ref, rtol=1e-7, atol=1e-9)
            assert np.all(dense[M:, :] == 0) and np.all(dense[:, M:] == 0)
        for npts in (1, 16, 37):
            X = rng.standard_normal((npts, w.D)) * 1.5
            fm, fv = pack.predict(torch.tensor(X, device=DEV))
            fm_ref, fv_ref = gp.predict(X)
            np.testing.assert_allclose(fm.cpu().numpy(), fm_ref, rtol=1e-8, atol=1e-11)
            np.testing.assert_allclose(fv.cpu().numpy(), fv_ref, rtol=1e-8, atol=1e-11)


@pytest.mark.parametrize('name', ['tiny', 'mini_sarcos', 'mini_smallscale'])
@pytest.mark.parametrize('cond', [True, False])
def test_elbo_forward_tri_form_matches_golden(tri, name, cond):
    from test_hip_parity import _load_golden, _compare
    z, w, p, noise = _load_golden(name)
    eng = ops.HipElbo(w.model_config(), DEV)
    eng.prepare(p)
    assert eng.gp_form() == 'tri/tri'
    ws = eng.run(z['u'], z['y'], noise, condition=cond)
    tag = 'c1_' if cond else 'c0_'
    _compare(ws, w, {k[3:]: z[k] for k in z.files if k.startswith(tag)})


@pytest.mark.parametrize('kw', [
    dict(M=100, dim_x=14, dim_u=7, dim_y=7, T=20, B=2, S=20, recog_len=4, k_factor=50., var_y=0.05 ** 2),   # Sarcos tile
    dict(M=130, dim_x=9, dim_u=3, dim_y=2, T=11, B=2, S=9, recog_len=3, k_factor=5.),                        # 10 waves, streamed
    dict(M=200, dim_x=14, dim_u=7, dim_y=7, T=9, B=1, S=20, recog_len=2, k_factor=50.),                     # C4 tile
    dict(M=250, dim_x=4, dim_u=2, dim_y=2, T=9, B=3, S=11, recog_len=3, k_factor=1.),                       # NBLK = 16
    dict(M=300, dim_x=4, dim_u=2, dim_y=2, T=9, B=2, S=9, recog_len=50, k_factor=1.),                       # C5 tile, T < R
    dict(M=20, dim_x=4, dim_u=1, dim_y=1, T=50, B=3, S=50, recog_len=16, k_factor=100., gp_len=2.),         # C1 tile
    dict(M=50, dim_x=4, dim_u=1, dim_y=1, T=33, B=1, S=1, recog_len=16, k_factor=100.),                     # single chain
])
def test_elbo_and_gradient_tri_form_match_oracle(tri, kw):
    """forward evaluation in the two-triangular form, and the adjoint fed by the A2 tiles it saved"""
    from test_hip_parity import _compare
    from oracle import cbfssm_torch_ref as tref
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
    out1, x1 = ws.out.clone(), ws.x.clone()
    ws = eng.run(u, y, noise, condition=True)
    assert torch.equal(out1, ws.out) and torch.equal(x1, ws.x)        # the flag hand-off does not change the arithmetic
    g = train.HipElboGrad(cfg, DEV)
    loss, grads, _ = g.loss_and_grads({k: torch.tensor(v, device=DEV) for k, v in p.items()}, u, y, noise)
    assert g.pack_f.gp_form() == 'tri'
    scal, gref = tref.loss_and_grads(cfg, p, u, y, noise, True)
    assert abs(float(loss) - scal['loss']) <= 1e-9 * abs(scal['loss'])
    for k in train.PARAM_NAMES:
        np.testing.assert_allclose(grads[k].cpu().numpy(), gref[k], rtol=1e-6, atol=1e-7 * np.abs(gref[k]).max())


def test_condition_estimate_and_automatic_form(monkeypatch):
    """scal[COND] = |K|_inf |K^-1|_inf of K_mm + jitter I; auto mode switches to the two-triangular form above the threshold"""
    monkeypatch.setenv('CBFSSM_GP_FORM', 'auto')
    orc = _oracle()
    w = dataclasses.replace(syn.WORKLOADS['C3'], B=2, T=12)
    for ls_mult, expect in ((1.0, 'dense'), (8.0, 'dense'), (32.0, 'dense'), (64.0, 'tri'), (128.0, 'tri')):
        p = syn.trained_like_params(w, ls_mult=ls_mult, zeta_mean=0.1)
        eng = ops.HipElbo(w.model_config(), DEV)
        eng.prepare(p)
        for g, pack in (('f', eng.pack_f), ('b', eng.pack_b)):
            gp = orc.GPModel(p[g + '.zeta_pos'], p[g + '.zeta_mean'], p[g + '.zeta_var_unc'], p[g + '.variance_unc'],
                             p[g + '.lengthscales_unc'])
            K = gp.kern.K(gp.zeta_pos) + 1e-8 * np.eye(w.M)
            cond_inf = np.abs(K).sum(1).max() * np.abs(np.linalg.inv(K)).sum(1).max()
            got = float(pack.scal[lib.SCAL_COND])
            assert got == pytest.approx(cond_inf, rel=1e-3 * max(1.0, cond_inf * 1e-10)), (ls_mult, got, cond_inf)
            assert pack.gp_form() == expect, (ls_mult, got, pack.gp_form())
    # a train step follows the parameters: the read-back lags by a step, the form flips without anybody waiting for it
    p0 = syn.trained_like_params(w, ls_mult=4.0, zeta_mean=0.1)
    st = train.HipTrainStep(w.model_config(), {k: torch.tensor(v, device=DEV) for k, v in p0.items()}, DEV)
    u, y = syn.make_inputs(w)
    noise = syn.make_noise(w)
    st.step(u, y, noise)
    assert st.engine.pack_f.gp_form() == 'dense'
    big = syn.trained_like_params(w, ls_mult=64.0, zeta_mean=0.1)
    for k in train.PARAM_NAMES:
        st.params[k].copy_(torch.tensor(big[k], device=DEV))
    for _ in range(4):
        st.step(u, y, noise)
        torch.cuda.synchronize()
    assert st.engine.pack_f.gp_form() == 'tri' and st.engine.pack_b.gp_form() == 'tri'
