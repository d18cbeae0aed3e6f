"""cbfssm.model.gp_tf on the MI355X HIP path: the GP primitives of the reference's cbfssm/model/gp_tf.py with the same
names and argument meaning -- `RBF`, `cast_cholesky`, `conditional`, `GPModel.predict / prior_kl` -- evaluated by the
library behind include/cbfssm_hip.h.  Inputs are numpy arrays or torch tensors; results are float64 torch tensors on the
device.  (The model classes do not go through this module: they drive the fused pass kernels directly.  This is the
stand-alone surface of the same kernels, for callers that hold GP objects rather than a whole model.)

    kern = RBF(variance, lengthscales)                 gp_tf.py:20-49
    kern.K(X), kern.K(X, X2), kern.Kdiag(X)
    cast_cholesky(mat, jitter=1e-8)                    gp_tf.py:52-65  (float64 whatever the input dtype)
    conditional(Xnew, X, kern, f, q_sqrt, Lm=None)     gp_tf.py:68-100 (q_sqrt None, (M, Do) or (Do, M, M))
    GPModel(in_dim, out_dim, num_points, gp_var, gp_len, zeta_mean, zeta_pos, zeta_var)      gp_tf.py:103-172
"""
import ctypes as C
import numpy as np
import torch

from ..hip import lib as _l
from ..hip import ops
from ..hip.ops import _f64, _ptr, _stream
from ..synthetic import softplus_inverse
from .session import InvalidArgumentError


def _device(device=None):
    if device is not None:
        return torch.device(device)
    if not torch.cuda.is_available():
        raise RuntimeError('cbfssm needs an MI355X: no HIP device is visible and there is no CPU fallback')
    return torch.device('cuda', torch.cuda.current_device())


def forward(x):
    """softplus(x) + 1e-10 (tf_transform.py:19-21)"""
    return ops.tf_forward(x)


def backward(y):
    """inverse of forward on the host (tf_transform.py:13-16)"""
    return softplus_inverse(y)


class RBF:
    """ARD squared-exponential kernel, parameters kept unconstrained like the reference's tf.Variables."""

    def __init__(self, variance, lengthscales, dtype='float64', device=None):
        self.device = _device(device)
        self.variance_unc = torch.tensor(np.atleast_1d(backward(variance)), dtype=torch.float64, device=self.device)
        self.lengthscales_unc = torch.tensor(np.atleast_1d(backward(lengthscales)), dtype=torch.float64, device=self.device)

    @property
    def variance(self):
        return forward(self.variance_unc)

    @property
    def lengthscales(self):
        return forward(self.lengthscales_unc)

    def Kdiag(self, X):
        X = _f64(X, self.device)
        return self.variance.reshape(()).expand(X.shape[0]).clone()

    def K(self, X, X2=None):
        X = _f64(X, self.device)
        X2 = X if X2 is None else _f64(X2, self.device)
        assert X.dim() == 2 and X2.dim() == 2 and X.shape[1] == X2.shape[1] == self.lengthscales.numel()
        out = torch.empty(X.shape[0], X2.shape[0], dtype=torch.float64, device=self.device)
        ls, var = self.lengthscales.contiguous(), self.variance.contiguous()     # (named: they must outlive the launch)
        rc = _l.load().cbfssm_rbf_k_f64(X.shape[0], X2.shape[0], X.shape[1], _ptr(X), _ptr(X2), _ptr(ls), _ptr(var),
                                        _ptr(out), _stream())
        _l.check(rc, 'cbfssm_rbf_k_f64')
        return out


def cast_cholesky(mat, jitter=1e-8):
    """Lower Cholesky factor of mat + jitter I, computed in float64 whatever dtype comes in (gp_tf.py:57-65); a matrix
    that is not positive definite raises InvalidArgumentError as tf.cholesky does."""
    dev = mat.device if torch.is_tensor(mat) and mat.is_cuda else _device()
    in_dtype = mat.dtype if torch.is_tensor(mat) else None
    m = _f64(mat, dev)
    assert m.dim() == 2 and m.shape[0] == m.shape[1]
    M = m.shape[0]
    L = torch.empty_like(m)
    info = torch.zeros(1, dtype=torch.float64, device=dev)
    work = torch.empty(M * (M + 1) + 64, dtype=torch.float64, device=dev)
    rc = _l.load().cbfssm_cholesky_f64(M, _ptr(m), float(jitter), _ptr(L), _ptr(info), _ptr(work), _stream())
    _l.check(rc, 'cbfssm_cholesky_f64')
    if float(info[0]) != 0.0:
        raise InvalidArgumentError('Cholesky decomposition was not successful: leading minor %d is not positive definite'
                                   % int(info[0]))
    return L.to(in_dtype) if in_dtype is not None and in_dtype != torch.float64 else L


def _pack_for(kern, X, f, zeta_var):
    M, D = X.shape
    Do = f.shape[1]
    pack = ops.GPPack(M, D, Do, kern.device)
    pack.prepare(X, kern.lengthscales, kern.variance, f, zeta_var)
    if float(pack.scal[_l.SCAL_INFO]) != 0.0:
        raise InvalidArgumentError('Cholesky decomposition was not successful: leading minor %d of K(X) + 1e-8 I is not '
                                   'positive definite' % int(pack.scal[_l.SCAL_INFO]))
    return pack


def conditional(Xnew, X, kern, f, q_sqrt, Lm=None):
    """GPflow-1.0-style conditional (gp_tf.py:68-100): p(f* | q(f) = N(f, q_sqrt q_sqrt^T)), unwhitened.
    q_sqrt: None, (M, Do) standard deviations, or (Do, M, M) lower-triangular factors.  `Lm` is accepted for signature
    compatibility; the factor is recomputed from (X, kern) on the device (it is what Lm must equal, gp_tf.py:71-72).
    Returns (fmean (N, Do), fvar (N, Do))."""
    dev = kern.device
    X, Xnew, f = _f64(X, dev), _f64(Xnew, dev), _f64(f, dev)
    M, Do = f.shape
    if q_sqrt is None:
        pack = _pack_for(kern, X, f, torch.zeros(M, Do, dtype=torch.float64, device=dev))
        return pack.predict(Xnew)
    q = _f64(q_sqrt, dev).contiguous()
    if q.dim() == 2:
        assert q.shape == (M, Do)
        return _pack_for(kern, X, f, q * q).predict(Xnew)
    if q.dim() != 3:
        raise ValueError("bad dimension for q_sqrt")
    assert q.shape == (Do, M, M)
    pack = _pack_for(kern, X, f, torch.zeros(M, Do, dtype=torch.float64, device=dev))
    lib = _l.load()
    n = Xnew.shape[0]
    fmean = torch.empty(n, Do, dtype=torch.float64, device=dev)
    fvar = torch.empty_like(fmean)
    work = torch.empty(int(lib.cbfssm_gp_predict_fullq_work_elems(C.byref(pack.layout), n)), dtype=torch.float64, device=dev)
    rc = lib.cbfssm_gp_predict_fullq_f64(C.byref(pack.layout), _ptr(pack.buf), _ptr(q), _ptr(Xnew), n,
                                         _ptr(fmean), _ptr(fvar), _ptr(work), _stream())
    _l.check(rc, 'cbfssm_gp_predict_fullq_f64')
    return fmean, fvar


class GPModel:
    """Sparse GP with diagonal q(z) (gp_tf.py:103-172): inducing inputs / means drawn as the reference draws them
    (unseeded unless `seed` is given), `predict(Xnew)` and `prior_kl()` on the device."""

    def __init__(self, in_dim, out_dim, num_points, gp_var, gp_len, zeta_mean, zeta_pos, zeta_var, dtype='float64',
                 device=None, seed=None):
        self.in_dim, self.out_dim, self.num_points, self.dtype = in_dim, out_dim, num_points, dtype
        rng = np.random.default_rng(seed)
        dev = _device(device)
        self.zeta_pos = torch.tensor(rng.uniform(-zeta_pos, zeta_pos, size=(num_points, in_dim)), device=dev)   # :112-115
        self.zeta_mean = torch.tensor(zeta_mean * rng.random((num_points, out_dim)), device=dev)                 # :117-118
        self.zeta_var_unc = torch.tensor(backward(zeta_var * np.ones((num_points, out_dim))), device=dev)       # :120-121
        self.kern = RBF(gp_var, np.asarray([gp_len] * in_dim, dtype=np.float64), device=dev)                     # :125-127
        self._pack = ops.GPPack(num_points, in_dim, out_dim, dev)

    @property
    def zeta_var(self):
        return forward(self.zeta_var_unc)

    @property
    def zeta_std(self):
        return torch.sqrt(self.zeta_var)

    def _prepared(self):
        self._pack.prepare(self.zeta_pos, self.kern.lengthscales, self.kern.variance, self.zeta_mean, self.zeta_var)
        return self._pack

    @property
    def cholesky(self):
        return self._prepared().L.clone()                                                                        # :129-130

    def predict(self, Xnew):
        return self._prepared().predict(Xnew)                                                                    # :132-161

    def prior_kl(self):
        return self._prepared().scal[_l.SCAL_KLZ].clone()                                                        # :163-172
