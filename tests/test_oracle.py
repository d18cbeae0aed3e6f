"""Pins for the CPU oracle (SURVEY.md section 8c list): the reference has no tests or golden vectors of its own, so
the restatement is checked against independent implementations and closed-form answers that follow from the
reference code.  CPU only."""
import numpy as np
import pytest
import scipy.linalg as sla
import torch

from cbfssm import synthetic as syn
from oracle import cbfssm_oracle as orc
from oracle import cbfssm_torch_ref as tref


def _setup(w, perturb=True):
    p = syn.make_params(w, seed=1)
    if perturb:
        p = syn.perturb_params(p)
    u, y = syn.make_inputs(w, seed=0)
    noise = syn.make_noise(w, seed=2)
    return w.model_config(), p, u, y, noise


# (1) tf_transform round trip and the y>35 branch (tf_transform.py:13-21)
def test_softplus_roundtrip_and_large_branch():
    y = np.array([1e-9, 1e-4, 0.3, 1.0, 20.0, 34.9, 35.1, 80.0, 700.0])
    x = orc.tf_backward(y)
    np.testing.assert_allclose(orc.tf_forward(x), y, rtol=1e-9, atol=1e-15)
    assert x[-1] == pytest.approx(700.0 - 1e-10)
    with pytest.raises(AssertionError):
        orc.tf_backward(np.array([1e-10]))
    np.testing.assert_allclose(syn.softplus_inverse(y), x)


# (2) predict at the inducing inputs ~ (mu_z, var_z) up to O(jitter); far away -> (0, sigma_k^2)
def test_predict_known_answers():
    w = syn.tiny(M=9)
    _, p, _, _, _ = _setup(w, perturb=False)
    gp = orc.GPModel(p['f.zeta_pos'], p['f.zeta_mean'], p['f.zeta_var_unc'], p['f.variance_unc'], p['f.lengthscales_unc'])
    fm, fv = gp.predict(gp.zeta_pos)
    np.testing.assert_allclose(fm, gp.zeta_mean, rtol=0, atol=1e-6)
    np.testing.assert_allclose(fv, gp.zeta_var, rtol=0, atol=1e-6)
    far = 1e3 * np.ones((2, w.D))
    fm, fv = gp.predict(far)
    np.testing.assert_allclose(fm, 0.0, atol=1e-300)
    np.testing.assert_allclose(fv, float(gp.kern.variance[0]), rtol=1e-14)


# (3) trsm form == contraction form K^-1
def test_predict_trsm_equals_contraction():
    w = syn.tiny()
    _, p, _, _, _ = _setup(w)
    gp = orc.GPModel(p['b.zeta_pos'], p['b.zeta_mean'], p['b.zeta_var_unc'], p['b.variance_unc'], p['b.lengthscales_unc'])
    X = np.random.default_rng(5).standard_normal((17, w.D))
    fm, fv = gp.predict(X)
    K = gp.kern.K(gp.zeta_pos) + orc.JITTER * np.eye(w.M)
    np.testing.assert_allclose(gp.cholesky @ gp.cholesky.T, K, rtol=1e-12, atol=1e-14)
    Kinv = np.linalg.inv(K)
    Kmn = gp.kern.K(gp.zeta_pos, X)
    Bm = Kinv @ Kmn
    fm2 = Kmn.T @ (Kinv @ gp.zeta_mean)
    fv2 = (float(gp.kern.variance[0]) - np.sum(Kmn * Bm, 0))[:, None] + (Bm * Bm).T @ gp.zeta_var
    np.testing.assert_allclose(fm, fm2, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(fv, fv2, rtol=1e-9, atol=1e-12)
    # direct (x-z)^2 distance agrees with the expansion the reference uses
    d2 = (((gp.zeta_pos[:, None, :] - X[None]) / gp.kern.lengthscales) ** 2).sum(-1)
    np.testing.assert_allclose(gp.kern.square_dist(gp.zeta_pos, X), d2, rtol=1e-10, atol=1e-12)


# (4) prior_kl closed form == torch.distributions (gp_tf.py:163-172)
def test_prior_kl_matches_torch_distributions():
    w = syn.tiny()
    _, p, _, _, _ = _setup(w)
    for g in 'fb':
        args = [p[g + k] for k in ('.zeta_pos', '.zeta_mean', '.zeta_var_unc', '.variance_unc', '.lengthscales_unc')]
        kl_np = orc.GPModel(*args).prior_kl()
        kl_t = tref.GPModel(*[torch.tensor(a) for a in args]).prior_kl()
        assert kl_np == pytest.approx(float(kl_t), rel=1e-10)


# (5) window schedule: every t written exactly once over the two runs (cbfssm.py:123-128)
@pytest.mark.parametrize('T,R', [(50, 16), (250, 16), (11, 3), (100, 50), (7, 16), (64, 16)])
def test_window_schedule(T, R):
    r0, w0 = orc.window_schedule(T, R, 0)
    r1, w1 = orc.window_schedule(T, R, 1)
    assert np.all(w0 ^ w1)
    assert list(np.nonzero(r0)[0]) == [t for t in range(T) if (t + 1) % (2 * R) == 0]
    assert list(np.nonzero(r1)[0]) == [t for t in range(T) if (t + R + 1) % (2 * R) == 0]
    for t in range(T):
        assert tref.window_flags(t, R, 0) == (bool(r0[t]), bool(w0[t]))
        assert tref.window_flags(t, R, 1) == (bool(r1[t]), bool(w1[t]))


# (6)-(11) structural known answers of the ELBO
def test_elbo_known_answers():
    w = syn.tiny(k_factor=1.0)
    cfg, p, u, y, noise = _setup(w)
    o = orc.CBFSSMOracle(cfg, p)
    tr = {}
    res = o.run(u, y, noise, condition=True, trace=tr)
    # (6) k_factor = 1: sig = fvar*var_y/(fvar+var_y)
    fvar = tr['f_fvar'][0]
    sig = fvar * o.var_y / (fvar + o.var_y)
    k = fvar / (fvar + o.var_y)
    fmean = tr['f_fmean'][0]
    mu = fmean + k * (res['y_tilde'][:, 1] - fmean)
    x1 = mu + noise['eps_f'][0][:, :, None] * np.sqrt(sig)
    np.testing.assert_allclose(res['x_final'][:, 1], x1, rtol=1e-12)
    # (8) loglik at t=0: y_final[:,0] == y[:,0]
    assert np.array_equal(res['x_final'][:, 0, :, :w.dim_y], np.tile(y[:, 0, None, :], (1, w.S, 1)))
    # (9) entropy = sum over written steps of 0.5*sum(1+log 2pi+log fvar)
    ent = 0.0
    for run in (0, 1):
        _, wr = orc.window_schedule(w.T, w.recog_len, run)
        for t in range(w.T):
            if wr[t]:
                ent += 0.5 * np.sum(1 + np.log(2 * np.pi) + np.log(tr['b_fvar'][(run, t)]))
    assert res['entropy'] == pytest.approx(ent, rel=1e-13)
    # (10) pred_var is the population variance + var_y[:dim_y]
    yf = res['x_final'][..., :w.dim_y]
    np.testing.assert_allclose(res['pred_var'], ((yf - yf.mean(2, keepdims=True)) ** 2).mean(2) + o.var_y[:w.dim_y])
    # (11) ELBO combination
    lf = cfg['loss_factors']
    elbo = (res['loglik'] - res['kl_x']) * lf[0] / w.S + res['entropy'] * lf[1] / w.S - res['kl_z_f'] - res['kl_z_b']
    assert res['loss'] == pytest.approx(-elbo, rel=1e-14)
    # y_tilde layout: first dim_y dims are the observations, the rest are written by exactly one run
    assert np.array_equal(res['y_tilde'][..., :w.dim_y], np.tile(y[:, :, None, :], (1, 1, w.S, 1)))
    for t in range(w.T):
        run = 0 if (t % (2 * w.recog_len)) < w.recog_len else 1
        np.testing.assert_array_equal(res['y_tilde'][:, t, :, w.dim_y:], tr['b_h'][(run, t)])


# (7) condition=False and t >= R-1: free run, zero KL (cbfssm.py:227-234)
def test_condition_false_free_runs():
    w = syn.tiny()
    cfg, p, u, y, noise = _setup(w)
    o = orc.CBFSSMOracle(cfg, p)
    tr = {}
    res = o.run(u, y, noise, condition=False, trace=tr)
    R = w.recog_len
    t = R - 1
    x_free = tr['f_fmean'][t] + noise['eps_f'][t][:, :, None] * np.sqrt(tr['f_fvar'][t])
    np.testing.assert_allclose(res['x_final'][:, t + 1], x_free, rtol=1e-13)
    res_c = o.run(u, y, noise, condition=True)
    assert res['kl_x'] < res_c['kl_x']
    np.testing.assert_allclose(res['x_final'][:, :R], res_c['x_final'][:, :R], rtol=1e-13)


# numpy oracle == torch restatement (two independent codings of the same reference lines)
@pytest.mark.parametrize('condition', [True, False])
def test_numpy_oracle_matches_torch_restatement(condition):
    w = syn.tiny()
    cfg, p, u, y, noise = _setup(w)
    res = orc.elbo_step(cfg, p, u, y, noise, condition)
    tp = {k: torch.tensor(v) for k, v in p.items()}
    tn = {k: torch.tensor(v) for k, v in noise.items()}
    out = tref.elbo_step(cfg, tp, torch.tensor(u), torch.tensor(y), tn, condition, want_pred=True)
    for k in ('loss', 'loglik', 'kl_x', 'entropy', 'kl_z_f', 'kl_z_b'):
        assert float(out[k]) == pytest.approx(res[k], rel=1e-10), k
    np.testing.assert_allclose(out['pred_mean'].numpy(), res['pred_mean'], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(out['pred_var'].numpy(), res['pred_var'], rtol=1e-10, atol=1e-14)
    np.testing.assert_allclose(out['x_final'].numpy(), res['x_final'], rtol=1e-10, atol=1e-12)


# (12) gradients: autograd of the torch restatement vs central finite differences of the numpy oracle
def test_gradients_vs_finite_differences():
    w = syn.tiny(T=7, B=2, S=3, M=6, recog_len=2)
    cfg, p, u, y, noise = _setup(w)
    _, grads = tref.loss_and_grads(cfg, p, u, y, noise, True)
    rng = np.random.default_rng(11)
    for name in syn.PARAM_NAMES:
        g = grads[name]
        flat = p[name].reshape(-1)
        for idx in rng.choice(flat.size, size=min(3, flat.size), replace=False):
            h = 1e-6 * max(1.0, abs(flat[idx]))
            pp = {k: v.copy() for k, v in p.items()}
            pm = {k: v.copy() for k, v in p.items()}
            pp[name].reshape(-1)[idx] += h
            pm[name].reshape(-1)[idx] -= h
            fd = (orc.elbo_step(cfg, pp, u, y, noise)['loss'] - orc.elbo_step(cfg, pm, u, y, noise)['loss']) / (2 * h)
            assert g.reshape(-1)[idx] == pytest.approx(fd, rel=2e-5, abs=1e-6), (name, idx)


# ---- CBFSSMHALF restatements (cbfssm/model/cbfssmhalf.py): two codings agree; gradient vs finite differences
def _half_setup(recog='rnn', **kw):
    w = syn.tiny(**kw)
    cfg = w.model_config()
    cfg['var_y'] = np.asarray([w.var_y] * w.dim_y)
    cfg['recog_model'] = recog
    rng = np.random.default_rng(4)
    p = {k: v for k, v in syn.perturb_params(syn.make_params(w, seed=1)).items() if k.startswith('f.') or k == 'var_x_unc'}
    p['var_y_unc'] = syn.softplus_inverse(cfg['var_y']) + 0.1 * rng.standard_normal(w.dim_y)
    if recog == 'rnn':
        n_in, H = w.dim_u + w.dim_y, 16
        p.update({'recog.gate_kernel': 0.3 * rng.standard_normal((n_in + H, 2 * H)), 'recog.gate_bias': np.ones(2 * H),
                  'recog.cand_kernel': 0.3 * rng.standard_normal((n_in + H, H)), 'recog.cand_bias': 0.1 * rng.standard_normal(H),
                  'recog.dense_kernel': 0.3 * rng.standard_normal((H, w.dim_x)), 'recog.dense_bias': 0.1 * rng.standard_normal(w.dim_x)})
    u, y = syn.make_inputs(w)
    noise = {'eps_f': syn.make_noise(w)['eps_f']}
    return w, cfg, p, u, y, noise


@pytest.mark.parametrize('recog', ['rnn', 'output'])
def test_half_numpy_matches_torch_and_fd(recog):
    w, cfg, p, u, y, noise = _half_setup(recog, T=6, B=2, S=3, M=6, recog_len=3)
    res = orc.CBFSSMHALFOracle(cfg, p).run(u, y, noise, True)
    loss, grads = tref.half_loss_and_grads(cfg, p, u, y, noise, True)
    assert loss == pytest.approx(res['loss'], rel=1e-10)
    # hidden dims get no Kalman update: x_{t+1}[d >= dim_y] = fmean + eps sqrt(fvar)  (cbfssmhalf.py:155-160)
    rng = np.random.default_rng(3)
    for name in p:
        flat = p[name].reshape(-1)
        for idx in rng.choice(flat.size, size=min(2, flat.size), replace=False):
            h = 1e-6 * max(1.0, abs(flat[idx]))
            pp = {k: v.copy() for k, v in p.items()}
            pm = {k: v.copy() for k, v in p.items()}
            pp[name].reshape(-1)[idx] += h
            pm[name].reshape(-1)[idx] -= h
            fd = (orc.CBFSSMHALFOracle(cfg, pp).run(u, y, noise)['loss']
                  - orc.CBFSSMHALFOracle(cfg, pm).run(u, y, noise)['loss']) / (2 * h)
            assert grads[name].reshape(-1)[idx] == pytest.approx(fd, rel=5e-5, abs=1e-6), (name, idx)


# ---- PR-SSM restatement (cbfssm/model/prssm.py): free run == CBFSSMHALF with conditioning off; gradient vs FD
def _prssm_setup(recog='rnn', **kw):
    w = syn.tiny(**kw)
    cfg = w.model_config()
    cfg['var_y'] = np.asarray([w.var_y] * w.dim_y)
    cfg['recog_model'] = recog
    rng = np.random.default_rng(6)
    base = syn.perturb_params(syn.make_params(w, seed=1))
    p = {k[2:]: v for k, v in base.items() if k.startswith('f.') and 'lengthscales' not in k}
    p['lengthscales_unc'] = syn.softplus_inverse(np.asarray([w.gp_len])) + 0.1
    p['var_x_unc'] = base['var_x_unc']
    p['var_y_unc'] = syn.softplus_inverse(cfg['var_y']) + 0.1 * rng.standard_normal(w.dim_y)
    n_in, H = w.dim_u + w.dim_y, 16
    if recog == 'rnn':
        p.update({'recog.gate_kernel': 0.3 * rng.standard_normal((n_in + H, 2 * H)), 'recog.gate_bias': np.ones(2 * H),
                  'recog.cand_kernel': 0.3 * rng.standard_normal((n_in + H, H)), 'recog.cand_bias': 0.1 * rng.standard_normal(H),
                  'recog.dense_kernel': 0.3 * rng.standard_normal((H, w.dim_x)), 'recog.dense_bias': 0.1 * rng.standard_normal(w.dim_x)})
    elif recog == 'conv':
        flat = 5 * ((w.recog_len - 2) // 2)
        p.update({'recog.conv_kernel': 0.4 * rng.standard_normal((3, n_in, 5)), 'recog.conv_bias': 0.1 * rng.standard_normal(5),
                  'recog.dense_kernel': 0.3 * rng.standard_normal((flat, w.dim_x)), 'recog.dense_bias': 0.1 * rng.standard_normal(w.dim_x)})
    u, y = syn.make_inputs(w)
    noise = {'eps_f': syn.make_noise(w)['eps_f']}
    return w, cfg, p, u, y, noise


def test_prssm_is_half_without_conditioning_and_fd():
    w, cfg, p, u, y, noise = _prssm_setup('output', T=6, B=2, S=3, M=6, recog_len=3)
    res, grads = tref.prssm_loss_and_grads(cfg, p, u, y, noise)
    # the numpy CBFSSMHALF oracle with condition=False and recog_len=1 free-runs from the same x_0
    hp = {'f.' + k: v for k, v in p.items() if k in ('zeta_pos', 'zeta_mean', 'zeta_var_unc', 'variance_unc')}
    hp['f.lengthscales_unc'] = np.tile(p['lengthscales_unc'], w.D)
    hp['var_x_unc'], hp['var_y_unc'] = p['var_x_unc'], p['var_y_unc']
    hcfg = dict(cfg)
    hcfg['recog_len'] = 1
    half = orc.CBFSSMHALFOracle(hcfg, hp).run(u, y, noise, condition=False)
    np.testing.assert_allclose(res['x_final'], half['x_final'], rtol=1e-10, atol=1e-12)
    assert half['kl_x'] == 0.0
    assert float(res['loglik']) == pytest.approx(half['loglik'], rel=1e-10)
    # the prior KL without jitter differs from the jittered one only at the 1e-8 level here
    assert float(res['kl_z']) == pytest.approx(half['kl_z_f'], rel=1e-5)
    rng = np.random.default_rng(3)
    for name in p:
        flat = p[name].reshape(-1)
        for idx in rng.choice(flat.size, size=min(2, flat.size), replace=False):
            h = 1e-6 * max(1.0, abs(flat[idx]))
            pp = {k: v.copy() for k, v in p.items()}
            pm = {k: v.copy() for k, v in p.items()}
            pp[name].reshape(-1)[idx] += h
            pm[name].reshape(-1)[idx] -= h
            fd = (float(tref.prssm_loss_and_grads(cfg, pp, u, y, noise)[0]['loss'])
                  - float(tref.prssm_loss_and_grads(cfg, pm, u, y, noise)[0]['loss'])) / (2 * h)
            assert grads[name].reshape(-1)[idx] == pytest.approx(fd, rel=5e-5, abs=1e-6), (name, idx)


def test_trained_like_fixture_is_this_oracles_output():
    """tests/golden/trained_C3.npz (read by the GPU parity tests instead of recomputing minutes of oracle time): one sweep
    point re-derived here -- inputs from the seeds, outputs from the numpy oracle -- must reproduce the fixture."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle'))
    import make_golden as mg
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'trained_C3.npz'))
    m = 16
    w, p, u, y, noise = mg.trained_case('C3', m)
    tag = 'x%d_' % m
    assert abs(mg.param_checksum(p) - float(z[tag + 'param_checksum'])) <= 1e-9 * abs(float(z[tag + 'param_checksum']))
    ref = orc.elbo_step(w.model_config(), p, u, y, noise, True)
    assert abs(ref['loss'] - float(z[tag + 'loss'])) <= 1e-12 * abs(ref['loss'])
    np.testing.assert_allclose(ref['pred_mean'], z[tag + 'pred_mean'], rtol=0, atol=1e-9 * np.abs(ref['pred_mean']).max())
    np.testing.assert_allclose(ref['pred_var'], z[tag + 'pred_var'], rtol=1e-8)
    np.testing.assert_allclose(ref['x_final'][:, z['t_sel']], z[tag + 'x_final_sel'], rtol=0,
                               atol=1e-9 * np.abs(ref['x_final']).max())


# ---- the noise generator's restatement (oracle/philox.py): pinned by the Philox paper's known-answer vectors
def test_philox_known_answers():
    from oracle import philox as ph
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]            # Random123 kat_vectors, philox4x32 10 rounds
    for ctr, key, want in kat:
        got = ph.philox4x32_10(np.array(ctr, dtype=np.uint64), np.array(key, dtype=np.uint64))
        assert [int(v) for v in got] == list(want)
    # a draw is a function of (seed, offset + i): splitting it changes nothing; the moments are a standard normal's
    z = ph.normal(2024, 0, 400001)
    np.testing.assert_array_equal(np.concatenate([ph.normal(2024, 0, 1237), ph.normal(2024, 1237, 400001 - 1237)]), z)
    assert abs(z.mean()) < 5e-3 and abs(z.std() - 1.0) < 5e-3 and abs((z ** 3).mean()) < 2e-2 and abs((z ** 4).mean() - 3.0) < 5e-2
    assert np.abs(ph.normal(2025, 0, 1000) - z[:1000]).max() > 1.0                       # another seed, another stream
    assert abs(np.corrcoef(z[:-1], z[1:])[0, 1]) < 5e-3
