"""BASELINE.json's headline workload at FULL size (C3: Sarcos-shaped, M=100, T=250, B=256, S=20 -- the CPU oracle would
need minutes per evaluation there), checked through properties that do not depend on the size:

* two evaluations are bit-identical (fixed-order reductions, no atomics);
* the data terms are additive over the batch -- two half batches with their slices of the noise give the same
  loglik / kl_x / entropy as the whole batch, the prior KL counted once (what the data-parallel step relies on,
  cbfssm.py:257-261 sums over the batch);
* the loss is the reference's linear combination of its five terms for any loss_factors (cbfssm.py:257-261);
* sequences do not interact: permuting the batch permutes the predictive moments and leaves the loss unchanged;
* the analytic gradient of all twelve tensors agrees with a central finite difference of the loss along a random
  direction (the adjoint kernels against the forward kernels, no oracle involved).
"""
import numpy as np
import pytest
import torch

from cbfssm import synthetic as syn
from cbfssm.hip import train

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _workload(name):
    w = syn.WORKLOADS[name]
    p = syn.perturb_params(syn.make_params(w, seed=1), scale=0.1)
    u, y = syn.make_inputs(w)
    noise = syn.make_noise(w)
    params = {k: torch.tensor(v, device=DEV) for k, v in p.items()}
    return w, params, u, y, noise


@pytest.fixture(scope='module')
def c3():
    return _workload('C3')


def _terms(t):
    return {k: float(t[k]) for k in ('loglik', 'kl_x', 'entropy', 'kl_z_f', 'kl_z_b')}


def test_full_size_is_deterministic_and_additive_over_the_batch(c3):
    w, params, u, y, noise = c3
    eng = train.HipElboGrad(w.model_config(), DEV)
    # (forward() returns views into the engine's workspace: take the numbers before the next evaluation overwrites them)
    l0, t0, ws = eng.forward(params, u, y, noise)
    l0, full, pm0 = float(l0), _terms(t0), ws.pred_mean.clone()
    l1, t1, ws = eng.forward(params, u, y, noise)
    assert l0 == float(l1) and full == _terms(t1) and torch.equal(pm0, ws.pred_mean)
    h = w.B // 2
    parts = []
    for lo, hi in ((0, h), (h, w.B)):
        nz = {'hid_b': np.ascontiguousarray(noise['hid_b'][:, :, lo:hi]),
              'eps_b': np.ascontiguousarray(noise['eps_b'][:, :, lo:hi]),
              'eps_f': np.ascontiguousarray(noise['eps_f'][:, lo:hi])}
        _, t, ws_h = eng.forward(params, u[lo:hi], y[lo:hi], nz)
        parts.append(_terms(t))
        np.testing.assert_allclose(ws_h.pred_mean.cpu().numpy(), pm0[lo:hi].cpu().numpy(), rtol=1e-12, atol=1e-14)
    for k in ('loglik', 'kl_x', 'entropy'):
        assert parts[0][k] + parts[1][k] == pytest.approx(full[k], rel=1e-11)
    for k in ('kl_z_f', 'kl_z_b'):
        assert parts[0][k] == full[k] == parts[1][k]


def test_full_size_loss_is_the_linear_combination_of_its_terms(c3):
    w, params, u, y, noise = c3
    base = None
    for lf in ((6.0, 0.0), (1.0, 1.0), (0.5, 3.0)):
        cfg = dict(w.model_config())
        cfg['loss_factors'] = np.asarray(lf)
        loss, t, _ = train.HipElboGrad(cfg, DEV).forward(params, u, y, noise)
        t = _terms(t)
        if base is None:
            base = t
        assert t == base                                   # the terms themselves do not depend on the factors
        elbo = lf[0] * (t['loglik'] - t['kl_x']) / w.S + lf[1] * t['entropy'] / w.S - t['kl_z_f'] - t['kl_z_b']
        assert float(loss) == pytest.approx(-elbo, rel=1e-12)


def test_full_size_sequences_do_not_interact(c3):
    w, params, u, y, noise = c3
    eng = train.HipElboGrad(w.model_config(), DEV)
    l0, _, ws = eng.forward(params, u, y, noise)
    l0 = float(l0)
    pm0, pv0 = ws.pred_mean.cpu().numpy().copy(), ws.pred_var.cpu().numpy().copy()
    perm = np.random.default_rng(5).permutation(w.B)
    nz = {'hid_b': np.ascontiguousarray(noise['hid_b'][:, :, perm]), 'eps_b': np.ascontiguousarray(noise['eps_b'][:, :, perm]),
          'eps_f': np.ascontiguousarray(noise['eps_f'][:, perm])}
    l1, _, ws = eng.forward(params, u[perm], y[perm], nz)
    np.testing.assert_array_equal(ws.pred_mean.cpu().numpy(), pm0[perm])
    np.testing.assert_array_equal(ws.pred_var.cpu().numpy(), pv0[perm])
    assert float(l1) == pytest.approx(l0, rel=1e-11)                 # (the batch sum runs in another order)


@pytest.mark.parametrize('name', ['C3', 'C4'])     # C4: M = 200, the stash-mode adjoint
def test_full_size_gradient_matches_directional_finite_differences(name):
    w, params, u, y, noise = _workload(name)
    eng = train.HipElboGrad(w.model_config(), DEV)
    loss, grads, _ = eng.loss_and_grads(params, u, y, noise)
    loss, grads = float(loss), {k: v.clone() for k, v in grads.items()}
    g = torch.Generator(device=DEV)
    g.manual_seed(11)
    for pname in train.PARAM_NAMES:                # one random direction per tensor: every one of the twelve is checked
        r = torch.randn(params[pname].shape, dtype=torch.float64, device=DEV, generator=g)
        slope = float((grads[pname] * r).sum())
        # step: the predicted change of the loss is 1e-6 of the loss (far above the rounding noise of a sum over 1.8e7
        # terms, ~1e-13 relative), capped where the gradient is small
        h = min(1e-6 * abs(loss) / max(abs(slope), 1e-300), 1e-4)
        lp = float(eng.forward({k: (v + h * r if k == pname else v) for k, v in params.items()}, u, y, noise)[0])
        lm = float(eng.forward({k: (v - h * r if k == pname else v) for k, v in params.items()}, u, y, noise)[0])
        fd = (lp - lm) / (2 * h)
        assert abs(fd - slope) <= 1e-4 * abs(slope) + 1e-12 * abs(loss) / h, (pname, fd, slope, h)


# ---------------------------------------------------------------------------------------------------------------------
# C5 (RoboMove-shaped: M = 300, T = 1000, S = 50, recog_len = 50) on the GPU at its own size
# ---------------------------------------------------------------------------------------------------------------------
def test_c5_full_size_eval_is_deterministic_additive_and_permutation_equivariant():
    """one eval step at the full per-GPU size (B = 512: 25 600 chains x 1000 steps, M = 300 tiles of 20 row blocks)"""
    w, params, u, y, noise = _workload('C5')
    eng = train.HipElboGrad(w.model_config(), DEV, require_adjoint=False)
    l0, t0, ws = eng.forward(params, u, y, noise)
    l0, full, pm0, pv0 = float(l0), _terms(t0), ws.pred_mean.clone(), ws.pred_var.clone()
    assert float(t0['info']) == 0.0 and np.isfinite(l0)
    l1, t1, ws = eng.forward(params, u, y, noise)
    assert l0 == float(l1) and full == _terms(t1) and torch.equal(pm0, ws.pred_mean)      # bit-identical repeat
    # two half batches with their slices of the noise: the data terms add up, the prior KL counts once
    h = w.B // 2
    parts = []
    for lo, hi in ((0, h), (h, w.B)):
        nz = {'hid_b': np.ascontiguousarray(noise['hid_b'][:, :, lo:hi]),
              'eps_b': np.ascontiguousarray(noise['eps_b'][:, :, lo:hi]),
              'eps_f': np.ascontiguousarray(noise['eps_f'][:, lo:hi])}
        _, t, ws_h = eng.forward(params, u[lo:hi], y[lo:hi], nz)
        parts.append(_terms(t))
        np.testing.assert_allclose(ws_h.pred_mean.cpu().numpy(), pm0[lo:hi].cpu().numpy(), rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(ws_h.pred_var.cpu().numpy(), pv0[lo:hi].cpu().numpy(), rtol=1e-12, atol=1e-14)
    for k in ('loglik', 'kl_x', 'entropy'):
        assert parts[0][k] + parts[1][k] == pytest.approx(full[k], rel=1e-11)
    assert parts[0]['kl_z_f'] == full['kl_z_f'] and parts[0]['kl_z_b'] == full['kl_z_b']
    # free-running prediction (condition = False beyond the first recog_len - 1 steps, cbfssm.py:227) keeps kl_x smaller
    nz0 = {'hid_b': np.ascontiguousarray(noise['hid_b'][:, :, :h]), 'eps_b': np.ascontiguousarray(noise['eps_b'][:, :, :h]),
           'eps_f': np.ascontiguousarray(noise['eps_f'][:, :h])}
    lc, tc, _ = eng.forward(params, u[:h], y[:h], nz0, condition=False)
    assert 0.0 < float(tc['kl_x']) < parts[0]['kl_x']              # only the first recog_len - 1 steps carry a KL term
    assert float(tc['entropy']) == parts[0]['entropy']             # the backward runs do not see `condition`


def test_c5_long_recurrence_gradient_through_chunked_stash_launches():
    """T = 1000, M = 300 at reduced batch with a stash budget that forces MANY time-chunked adjoint launches (the carried
    state adjoint gx_carry crosses every chunk boundary, resample boundaries of both runs fall inside chunks): analytic
    gradient against central finite differences of the loss along one random direction per tensor."""
    import dataclasses
    w = dataclasses.replace(syn.WORKLOADS['C5'], B=4)
    cfg = dict(w.model_config())
    cfg['adjoint_stash_gib'] = 0.25                 # 200 chains: ~100 steps per launch -> about ten launches per direction
    p = syn.perturb_params(syn.make_params(w, seed=1), scale=0.1)
    params = {k: torch.tensor(v, device=DEV) for k, v in p.items()}
    u, y = syn.make_inputs(w)
    noise = syn.make_noise(w)
    eng = train.HipElboGrad(cfg, DEV)
    assert eng.stash
    loss, grads, _ = eng.loss_and_grads(params, u, y, noise)
    loss, grads = float(loss), {k: v.clone() for k, v in grads.items()}
    # the same gradient with one launch per direction (default budget): chunking must not change it beyond summation order
    eng1 = train.HipElboGrad(w.model_config(), DEV)
    loss1, grads1, _ = eng1.loss_and_grads(params, u, y, noise)
    assert float(loss1) == loss
    for k in train.PARAM_NAMES:
        np.testing.assert_allclose(grads[k].cpu().numpy(), grads1[k].cpu().numpy(), rtol=1e-9,
                                   atol=1e-11 * float(grads1[k].abs().max()))
    g = torch.Generator(device=DEV)
    g.manual_seed(13)
    for pname in train.PARAM_NAMES:
        r = torch.randn(params[pname].shape, dtype=torch.float64, device=DEV, generator=g)
        slope = float((grads[pname] * r).sum())
        h = min(1e-6 * abs(loss) / max(abs(slope), 1e-300), 1e-4)
        lp = float(eng.forward({k: (v + h * r if k == pname else v) for k, v in params.items()}, u, y, noise)[0])
        lm = float(eng.forward({k: (v - h * r if k == pname else v) for k, v in params.items()}, u, y, noise)[0])
        fd = (lp - lm) / (2 * h)
        assert abs(fd - slope) <= 1e-4 * abs(slope) + 1e-12 * abs(loss) / h, (pname, fd, slope, h)
