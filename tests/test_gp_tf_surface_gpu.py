"""cbfssm.model.gp_tf -- the reference's GP primitives by name (gp_tf.py:20-172) on the HIP path -- against the oracle:
RBF.K / Kdiag, cast_cholesky of a given matrix (incl. the float32-in / float64-inside rule and the non-PD error),
conditional() in its three q_sqrt branches (None, standard deviations, full lower-triangular factors), GPModel."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _orc():
    from oracle import cbfssm_oracle as orc
    return orc


@pytest.mark.parametrize('M,D', [(7, 3), (100, 21), (200, 6)])
def test_rbf_and_cast_cholesky(M, D):
    from cbfssm.model import gp_tf
    from cbfssm.model.session import InvalidArgumentError
    orc = _orc()
    rng = np.random.default_rng(M)
    X, X2 = rng.uniform(-2, 2, (M, D)), rng.standard_normal((37, D))
    var, ls = 0.3, rng.uniform(0.8, 2.0, D)
    kern = gp_tf.RBF(var, ls)
    kref = orc.RBF(orc.tf_backward(np.array([var])), orc.tf_backward(ls))
    np.testing.assert_allclose(kern.K(X).cpu().numpy(), kref.K(X), rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(kern.K(X, X2).cpu().numpy(), kref.K(X, X2), rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(kern.Kdiag(X2).cpu().numpy(), kref.Kdiag(X2), rtol=1e-14)
    K = kref.K(X)
    L = gp_tf.cast_cholesky(torch.tensor(K, device='cuda:0')).cpu().numpy()
    np.testing.assert_allclose(L, orc.cast_cholesky(K), rtol=1e-7, atol=1e-10)
    np.testing.assert_allclose(L @ L.T, K + 1e-8 * np.eye(M), rtol=1e-11, atol=1e-13)
    assert np.all(np.triu(L, 1) == 0.0)
    L32 = gp_tf.cast_cholesky(torch.tensor(K, dtype=torch.float32, device='cuda:0'))       # gp_tf.py:57-65
    assert L32.dtype == torch.float32
    np.testing.assert_allclose(L32.cpu().numpy(), orc.cast_cholesky(K.astype(np.float32).astype(np.float64)), rtol=2e-4,
                               atol=1e-5)
    with pytest.raises(InvalidArgumentError):
        gp_tf.cast_cholesky(torch.tensor(-np.eye(M), device='cuda:0'))


@pytest.mark.parametrize('M,D,Do', [(12, 4, 3), (100, 21, 14), (130, 6, 4), (300, 6, 4)])
def test_conditional_three_q_sqrt_branches(M, D, Do):
    from cbfssm.model import gp_tf
    orc = _orc()
    rng = np.random.default_rng(M + Do)
    X, Xnew = rng.uniform(-2, 2, (M, D)), rng.standard_normal((41, D)) * 1.5
    f = 0.5 * rng.standard_normal((M, Do))
    var, ls = 0.4, rng.uniform(1.0, 2.0, D)
    kern = gp_tf.RBF(var, ls)
    kref = orc.RBF(orc.tf_backward(np.array([var])), orc.tf_backward(ls))
    q2 = 0.1 * np.exp(rng.uniform(-1, 1, (M, Do)))
    q3 = np.tril(0.05 * rng.standard_normal((Do, M, M))) + 0.1 * np.eye(M)[None]
    for q in (None, q2, q3):
        fm, fv = gp_tf.conditional(Xnew, X, kern, f, q)
        fm_ref, fv_ref = orc.conditional(Xnew, X, kref, f, q)
        np.testing.assert_allclose(fm.cpu().numpy(), fm_ref, rtol=1e-8, atol=1e-11)
        np.testing.assert_allclose(fv.cpu().numpy(), fv_ref, rtol=1e-8, atol=1e-11)
    with pytest.raises(ValueError):
        gp_tf.conditional(Xnew, X, kern, f, np.zeros((2, 2, 2, 2)))


def test_gpmodel_predict_and_prior_kl():
    from cbfssm.model import gp_tf
    orc = _orc()
    gp = gp_tf.GPModel(in_dim=6, out_dim=4, num_points=50, gp_var=0.25, gp_len=1.5, zeta_mean=0.05, zeta_pos=2.0,
                       zeta_var=0.01 ** 2, seed=3)
    ref = orc.GPModel(gp.zeta_pos.cpu().numpy(), gp.zeta_mean.cpu().numpy(), gp.zeta_var_unc.cpu().numpy(),
                      gp.kern.variance_unc.cpu().numpy(), gp.kern.lengthscales_unc.cpu().numpy())
    X = np.random.default_rng(0).standard_normal((33, 6))
    fm, fv = gp.predict(X)
    fm_ref, fv_ref = ref.predict(X)
    np.testing.assert_allclose(fm.cpu().numpy(), fm_ref, rtol=1e-8, atol=1e-12)
    np.testing.assert_allclose(fv.cpu().numpy(), fv_ref, rtol=1e-8, atol=1e-12)
    assert float(gp.prior_kl()) == pytest.approx(ref.prior_kl(), rel=1e-9)
    np.testing.assert_allclose(gp.cholesky.cpu().numpy(), ref.cholesky, rtol=1e-7, atol=1e-10)
