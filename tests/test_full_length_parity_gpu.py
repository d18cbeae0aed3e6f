"""GPU parity in the regimes BASELINE.json's 1e-5 tolerance is about (through the C ABI, against the CPU oracle):

(a) FULL-LENGTH recurrences at reduced batch -- C2 (T=100), C3 / C4 (T=250) and C5 (T=1000, M=300, R=50) with the
    configured S, M and recog_len -- against the committed fixtures tests/golden/full_C*.npz (oracle/make_golden.py):
    the five loss terms, the trajectories, the predictive mean/variance and the gradient of all twelve tensors;
(b) a TRAINED-LIKE, ill-conditioned parameter family (cbfssm.synthetic.trained_like_params: long lengthscales, inducing
    means of order 0.1, inducing variances 3e-3..3e-2) swept over cond(K_mm + 1e-8 I) = 5e3 .. 2e9 at the full C3
    recurrence, and single GP conditionals at points NEAR the inducing inputs (where sigma^2 - k^T K^-1 k cancels).

What 1e-5 can mean in (b).  The reference's own arithmetic (two triangular solves against chol(K_mm + 1e-8 I),
gp_tf.py:137-145) carries errors of order cond * eps into fmean and the recurrence amplifies them over T steps; two
float64 codings of that SAME algorithm on different BLAS back-ends (the numpy/LAPACK oracle and the PyTorch
restatement, both CPU) therefore differ by a `floor` that is measured here next to every HIP number.  The tests ask
HIP-vs-oracle <= 1e-5 wherever floor <= 1e-6 and <= FLOOR_MULT (3) x floor above (nothing can be closer to the reference
than the reference is to itself; round 2 needed 20 x at cond 2e9 -- the library multiplied by an explicitly formed L^-1
whose LEFT residual was of order cond(L) eps; the prepare kernel now refines it in doubled precision, csrc/cbfssm_api.hip).
Every achieved error is printed and collected in gpurun_out/parity_report.json.
"""
import dataclasses
import json
import os
import numpy as np
import pytest
import torch

from cbfssm import synthetic as syn
from cbfssm.hip import ops, train

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')
DEV = 'cuda:0'
BOUND = 1e-5                                   # north_star: relative, on the ELBO and the predictive mean/variance
FLOOR_MULT = 3.0                               # allowed multiple of the CPU-vs-CPU reproducibility floor where it exceeds 1e-6
REPORT = {}


def _report(key, val):
    REPORT[key] = val
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, 'parity_report.json'), 'w') as f:
            json.dump(REPORT, f, indent=1, sort_keys=True)
    except OSError:
        pass


def _rel_scalar(a, b):
    return abs(float(a) - float(b)) / max(abs(float(b)), 1e-300)


def _rel_max(a, b):
    """max |a - b| / max |b|: error relative to the scale of the tensor"""
    a, b = np.asarray(a), np.asarray(b)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def _rel_elem(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return float((np.abs(a - b) / np.maximum(np.abs(b), 1e-300)).max())


def _load_full(name):
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    wk = {k[len('workload_'):]: z[k] for k in z.files if k.startswith('workload_')}
    kw = {k: (tuple(v.tolist()) if v.ndim else v.item()) for k, v in wk.items()}
    w = syn.Workload(name, **kw)
    p = {k[len('param_'):]: z[k] for k in z.files if k.startswith('param_')}
    noise = {k[len('noise_'):]: z[k] for k in z.files if k.startswith('noise_')}
    return z, w, p, noise


def _errors(ws, w, ref, tsel=None):
    """achieved relative errors of one evaluation against a reference dict (oracle output or fixture)"""
    out = ws.out.cpu().numpy()
    e = {'loss': _rel_scalar(out[6], ref['loss'])}
    for i, k in enumerate(('loglik', 'kl_x', 'entropy', 'kl_z_f', 'kl_z_b')):
        e[k] = abs(out[i] - float(ref[k])) / max(abs(float(ref[k])), 1e-9)
    x = ops.as_btsd(ws.x, w.B, w.S).cpu().numpy()
    y2 = ops.as_btsd(ws.y2, w.B, w.S).cpu().numpy()
    if tsel is not None:
        e['x_final'] = _rel_max(x[:, tsel], ref['x_final_sel'])
        e['y2'] = _rel_max(y2[:, tsel], ref['y2_sel'])
    else:
        e['x_final'] = _rel_max(x, ref['x_final'])
        e['y2'] = _rel_max(y2, ref['y_tilde'][..., w.dim_y:])
    e['pred_mean'] = _rel_max(ws.pred_mean.cpu().numpy(), ref['pred_mean'])
    e['pred_var'] = _rel_elem(ws.pred_var.cpu().numpy(), ref['pred_var'])
    return e, int(out[7])


# ---------------------------------------------------------------------------------------------------------------------
# (a) full-length recurrences against the committed fixtures
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('name', ['full_C2', 'full_C3', 'full_C4', 'full_C5'])
@pytest.mark.parametrize('cond', [True, False])
def test_full_length_forward_matches_golden(name, cond):
    z, w, p, noise = _load_full(name)
    eng = ops.HipElbo(w.model_config(), DEV)
    eng.prepare(p)
    ws = eng.run(z['u'], z['y'], noise, condition=cond)
    tag = 'c1_' if cond else 'c0_'
    ref = {k[3:]: z[k] for k in z.files if k.startswith(tag)}
    e, info = _errors(ws, w, ref, z['t_sel'])
    print('\n%s condition=%d (T=%d M=%d S=%d R=%d): ' % (name, cond, w.T, w.M, w.S, w.recog_len) +
          ' '.join('%s %.1e' % kv for kv in sorted(e.items())))
    _report('%s/cond%d' % (name, cond), e)
    assert info == 0
    # well-conditioned (run-script initial values): both paths are float64 and differ in summation order only; the
    # achieved errors sit orders of magnitude below the 1e-5 bound -- held to 1e-8 so that a regression shows
    assert max(e.values()) <= 1e-8 < BOUND, e


@pytest.mark.parametrize('name', ['full_C2', 'full_C3', 'full_C4', 'full_C5'])
def test_full_length_gradient_matches_golden(name):
    z, w, p, noise = _load_full(name)
    eng = train.HipElboGrad(w.model_config(), DEV)
    params = {k: torch.tensor(v, device=DEV) for k, v in p.items()}
    loss, grads, terms = eng.loss_and_grads(params, z['u'], z['y'], noise, condition=True)
    assert float(terms['info']) == 0.0
    e = {'loss': _rel_scalar(loss, z['c1_loss'])}
    for k in train.PARAM_NAMES:
        e[k] = _rel_max(grads[k].cpu().numpy(), z['c1_grad_' + k])
    print('\n%s gradient: ' % name + ' '.join('%s %.1e' % kv for kv in sorted(e.items())))
    _report(name + '/grad', e)
    assert e['loss'] <= 1e-9
    assert max(e.values()) <= 1e-6, e          # relative to the largest entry of each tensor


# ---------------------------------------------------------------------------------------------------------------------
# (b) trained-like, ill-conditioned parameters
# ---------------------------------------------------------------------------------------------------------------------
def _trained_fixture():
    """tests/golden/trained_C3.npz (oracle/make_golden.py trained): the oracle's outputs on the trained-like parameter
    family, the floor |torch-CPU restatement - numpy oracle| of the same two-triangular-solve algorithm (the
    reproducibility of the reference formulation itself across float64 BLAS back-ends) and the autograd gradients.
    Minutes of CPU time when computed here (365 of the suite's 500 s in round 2), so they are committed; the inputs are
    regenerated from the seeds and a parameter checksum guards the generators (tests/test_oracle.py re-derives one
    sweep point on the CPU)."""
    return np.load(os.path.join(GOLDEN, 'trained_C3.npz'))


def _param_checksum(p):
    return float(sum(float(np.sum(v * np.cos(np.arange(v.size).reshape(v.shape)))) for _, v in sorted(p.items())))


SWEEP = [8, 16, 32, 64, 128, 256]    # lengthscale multipliers -> cond(K_mm + 1e-8 I) 5e3, 1e5, 2e6, 3e7, 4e8, 2e9


@pytest.mark.parametrize('ls_mult', SWEEP)
def test_trained_like_sweep_full_recurrence(ls_mult):
    z = _trained_fixture()
    tag = 'x%d_' % ls_mult
    w = dataclasses.replace(syn.WORKLOADS['C3'], B=2)
    cfg = w.model_config()
    p = syn.trained_like_params(w, ls_mult=float(ls_mult), zeta_mean=0.1)
    assert abs(_param_checksum(p) - float(z[tag + 'param_checksum'])) <= 1e-9 * abs(float(z[tag + 'param_checksum']))
    cond_f, cond_b = float(z[tag + 'cond_f']), float(z[tag + 'cond_b'])
    u, y = syn.make_inputs(w, seed=0)
    noise = syn.make_noise(w, seed=2)
    ref = {k[len(tag):]: z[k] for k in z.files if k.startswith(tag)}
    floor = {k: float(ref['floor_' + k]) for k in ('loss', 'pred_mean', 'pred_var')}
    eng = ops.HipElbo(cfg, DEV)
    eng.prepare(p)
    ws = eng.run(u, y, noise, condition=True)
    e, info = _errors(ws, w, ref, z['t_sel'])
    assert info == 0
    print('\nC3 (B=2, T=250) lengthscales x%d: cond f %.1e b %.1e | HIP vs oracle: loss %.1e pred_mean %.1e pred_var %.1e '
          'x %.1e | floor (oracle vs torch-CPU, same algorithm): loss %.1e pred_mean %.1e pred_var %.1e'
          % (ls_mult, cond_f, cond_b, e['loss'], e['pred_mean'], e['pred_var'], e['x_final'], floor['loss'],
             floor['pred_mean'], floor['pred_var']))
    _report('sweep_C3/ls_x%d' % ls_mult, {'cond_f': cond_f, 'cond_b': cond_b, 'hip': e, 'floor': floor,
                                          'gp_form': eng.gp_form() if hasattr(eng, 'gp_form') else 'dense'})
    for k in ('loss', 'pred_mean', 'pred_var'):
        lim = BOUND if floor[k] <= 1e-6 else max(BOUND, FLOOR_MULT * floor[k])
        assert e[k] <= lim, (k, e[k], lim, floor[k])


@pytest.mark.parametrize('ls_mult', [128, 256])
def test_refined_inverse_factor_left_residual(ls_mult):
    """The pack's G = L^-T (what the two-triangular kernels multiply by, gp_tf.py:137,145 being two substitutions in the
    reference) after the prepare kernel's Newton step in doubled precision: W = G^T must invert the pack's own L from the
    LEFT to rounding level, |W L - I| <= 4 eps max|W| max|L| -- a column-by-column substitution (what the elimination
    computes, reproduced here with scipy on the same L) leaves 20 x that at cond 2e9."""
    import scipy.linalg as sla
    w = dataclasses.replace(syn.WORKLOADS['C3'], B=2)
    p = syn.trained_like_params(w, ls_mult=float(ls_mult), zeta_mean=0.1)
    eng = ops.HipElbo(w.model_config(), DEV)
    eng.prepare(p)
    LD = np.longdouble
    eps = np.finfo(np.float64).eps
    for pk in (eng.pack_f, eng.pack_b):
        L = pk.L.cpu().numpy()
        W = pk.section('Linvt', (pk.M, pk.M)).cpu().numpy().T
        eye = np.eye(pk.M)
        left = float(np.abs(W.astype(LD) @ L.astype(LD) - eye).max())
        right = float(np.abs(L.astype(LD) @ W.astype(LD) - eye).max())
        W0 = sla.solve_triangular(L, eye, lower=True)
        left0 = float(np.abs(W0.astype(LD) @ L.astype(LD) - eye).max())
        lim = 4.0 * eps * np.abs(W).max() * np.abs(L).max()
        print('\nls x%d M=%d: |W L - I| %.2e (substitution on the same L: %.2e), |L W - I| %.2e, bound %.2e'
              % (ls_mult, pk.M, left, left0, right, lim))
        _report('refine/ls_x%d_Do%d' % (ls_mult, pk.Do), {'left': left, 'left_substitution': left0, 'right': right})
        assert left <= lim and right <= lim and left <= 0.25 * left0, (left, left0, right, lim)
        # K^-1 of the pack is G G^T of the refined factor
        Kinv = pk.Kinv.cpu().numpy()
        ref = (W.T.astype(LD) @ W.astype(LD)).astype(np.float64)
        assert np.abs(Kinv - ref).max() <= 1e-13 * np.abs(ref).max()


@pytest.mark.parametrize('M,D,Do,ls,spread', [(100, 21, 14, 8., 2.0), (100, 21, 14, 8., 1.0), (100, 21, 14, 16., 1.0),
                                             (100, 21, 14, 16., 0.5), (300, 6, 4, 2., 2.0), (300, 6, 4, 3., 2.0),
                                             (300, 6, 4, 4., 2.0), (200, 21, 14, 16., 0.7)])
def test_gp_predict_near_inducing_points(M, D, Do, ls, spread):
    """One GP conditional (cbfssm_gp_predict_f64, gp_tf.py:132-161) at points 0.1 lengthscale-units away from inducing
    inputs: fvar_0 = sigma^2 - |L^-1 k|^2 is 1e-4..1e-3 of sigma^2 there, so whatever is lost in the subtraction shows
    in fvar.  Reference value: the oracle's two triangular solves; the bound is 1e-5 of (fvar + var_x), var_x = 4e-6
    (run_sarcos.py), next to the error of the solves themselves against extended precision."""
    from oracle import cbfssm_oracle as orc
    rng = np.random.default_rng(M + int(ls))
    Z = rng.uniform(-spread, spread, (M, D))
    mu = 0.5 * rng.standard_normal((M, Do))
    s2 = 1e-4 * np.exp(rng.uniform(-1, 1, (M, Do)))
    lsv = np.full(D, ls)
    var = np.array([0.25])
    X = Z[rng.integers(0, M, 512)] + 0.1 * ls * 0.1 * rng.standard_normal((512, D))
    gp = orc.GPModel(Z, mu, orc.tf_backward(s2), orc.tf_backward(var), orc.tf_backward(lsv))
    fm_ref, fv_ref = gp.predict(X)
    K = gp.kern.K(Z) + 1e-8 * np.eye(M)
    cond = float(np.linalg.cond(K))
    pack = ops.GPPack(M, D, Do, DEV).prepare(torch.tensor(Z, device=DEV), torch.tensor(lsv, device=DEV),
                                             torch.tensor(var, device=DEV), torch.tensor(mu, device=DEV),
                                             torch.tensor(s2, device=DEV))
    fm, fv = pack.predict(torch.tensor(X, device=DEV))
    var_x = 4e-6
    e_m = _rel_max(fm.cpu().numpy(), fm_ref)
    e_v = float((np.abs(fv.cpu().numpy() - fv_ref) / (fv_ref + var_x)).max())
    print('\ngp_predict M=%d D=%d ls=%g spread=%g: cond %.1e, fvar median %.1e min %.1e | fmean %.1e fvar %.1e (of fvar+var_x)'
          % (M, D, ls, spread, cond, np.median(fv_ref), fv_ref.min(), e_m, e_v))
    _report('gp_near/M%d_ls%g_s%g' % (M, ls, spread), {'cond': cond, 'fmean': e_m, 'fvar': e_v,
                                                      'gp_form': pack.gp_form() if hasattr(pack, 'gp_form') else 'dense'})
    assert e_m <= BOUND and e_v <= BOUND, (cond, e_m, e_v)


@pytest.mark.parametrize('ls_mult,base,T', [(64, 'C3', 250), (32, 'C4', 60)])      # (x16 at C3: 7e-10)
def test_trained_like_gradient_full_recurrence(ls_mult, base, T):
    """The adjoint on ill-conditioned K_mm (two-triangular forward form chosen automatically, dense K^-1-adjoint
    accumulation fed by its saved A2 tiles) against reverse-mode autodiff of the float64 restatement through the
    reference's two triangular solves, full C3 recurrence (and the stash-mode adjoint of the C4 tile at T = 60).  The
    kernel-variance entry is the sensitive one: d loss / d sigma^2 = tr(Kbar K_mm) / sigma^2 + ..., and the entry sum of
    Kbar o K_mm cancels numbers of order cond^2 (3.5e-2 off at cond 3e7 before the train tail used
    tr(Kbar K_mm) = -tr(T) + jitter tr(T K^-1) + 0.5 Do (M - jitter tr K^-1), csrc/cbfssm_tail.hip)."""
    z = _trained_fixture()
    tag = 'g_%s_x%d_' % (base, ls_mult)
    w = dataclasses.replace(syn.WORKLOADS[base], B=2, T=T)
    cfg = w.model_config()
    p = syn.trained_like_params(w, ls_mult=float(ls_mult), zeta_mean=0.1)
    assert abs(_param_checksum(p) - float(z[tag + 'param_checksum'])) <= 1e-9 * abs(float(z[tag + 'param_checksum']))
    u, y = syn.make_inputs(w, seed=0)
    noise = syn.make_noise(w, seed=2)
    scal = {'loss': float(z[tag + 'loss'])}
    gref = {k: z[tag + 'grad_' + k] for k in train.PARAM_NAMES}
    eng = train.HipElboGrad(cfg, DEV)
    loss, grads, terms = eng.loss_and_grads({k: torch.tensor(v, device=DEV) for k, v in p.items()}, u, y, noise)
    assert float(terms['info']) == 0.0
    e = {'loss': _rel_scalar(loss, scal['loss'])}
    for k in train.PARAM_NAMES:
        e[k] = _rel_max(grads[k].cpu().numpy(), gref[k])
    print('\n%s (B=2, T=%d) lengthscales x%d gradient (form %s): ' % (base, T, ls_mult, eng.pack_f.gp_form()) +
          ' '.join('%s %.1e' % kv for kv in sorted(e.items())))
    _report('sweep_%s_grad/ls_x%d' % (base, ls_mult), e)
    assert e['loss'] <= 1e-7
    assert max(e.values()) <= 1e-4, e          # relative to the largest entry of each tensor
