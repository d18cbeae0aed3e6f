"""Synthetic CBF-SSM workloads (shapes, parameter initialisation, noise) shared by tests, bench and fixtures.

The reference never seeds its RNG and draws noise inside the graph (cbfssm/model/cbfssm.py:134,149,209), so
for any comparison the noise tensors and the initial inducing parameters are explicit inputs here:

    hid_b  (2, T, B, S)   hidden-state resample draws of the two backward runs   (cbfssm.py:133-136)
    eps_b  (2, T, B, S)   reparameterisation noise of the two backward runs      (cbfssm.py:149)
    eps_f  (T-1, B, S)    reparameterisation noise of the forward pass           (cbfssm.py:209)

One normal per (b, s) and step, broadcast over the state dimensions, exactly as the reference tiles it.

Parameter initialisation follows cbfssm/model/gp_tf.py:104-127 and cbfssm/model/cbfssm.py:30-54: the twelve
trainable tensors are kept *unconstrained* (inverse-softplus of the configured positive value) like the
reference's tf.Variables.
"""
from dataclasses import dataclass, field, asdict
import numpy as np

PARAM_NAMES = (
    'f.zeta_pos', 'f.zeta_mean', 'f.zeta_var_unc', 'f.variance_unc', 'f.lengthscales_unc',
    'b.zeta_pos', 'b.zeta_mean', 'b.zeta_var_unc', 'b.variance_unc', 'b.lengthscales_unc',
    'var_x_unc', 'var_y_unc',
)


@dataclass
class Workload:
    name: str
    dim_u: int
    dim_y: int
    dim_x: int
    M: int            # ind_pnt_num
    T: int            # seq_len
    B: int            # batch_size (per GPU)
    S: int            # samples (particles)
    recog_len: int
    k_factor: float
    loss_factors: tuple = (1.0, 0.0)
    zeta_pos: float = 2.0
    zeta_mean: float = 0.05 ** 2
    zeta_var: float = 0.01 ** 2
    var_x: float = 0.002 ** 2
    var_y: float = 1.0
    gp_var: float = 0.5 ** 2
    gp_len: float = 1.0
    learning_rate: float = 0.01

    @property
    def D(self):
        return self.dim_x + self.dim_u

    @property
    def dim_out_b(self):
        return self.dim_x - self.dim_y

    @property
    def N(self):
        return self.B * self.S

    def model_config(self, ds_cls=None):
        """The dict the reference's run scripts hand to the model constructor (run/template.py:19-40)."""
        if ds_cls is None:
            ds_cls = type('SyntheticDS', (), {'dim_u': self.dim_u, 'dim_y': self.dim_y})
        return {
            'ds': ds_cls, 'batch_size': self.B, 'shuffle': 10000,
            'dim_x': self.dim_x, 'ind_pnt_num': self.M, 'samples': self.S,
            'learning_rate': self.learning_rate, 'loss_factors': np.asarray(self.loss_factors, dtype=np.float64),
            'k_factor': self.k_factor, 'recog_len': self.recog_len,
            'zeta_pos': self.zeta_pos, 'zeta_mean': self.zeta_mean, 'zeta_var': self.zeta_var,
            'var_x': np.asarray([self.var_x] * self.dim_x), 'var_y': np.asarray([self.var_y] * self.dim_x),
            'gp_var': self.gp_var, 'gp_len': self.gp_len,
        }

    def flops_per_state(self):
        """Algorithmic FLOPs per (sequence, timestep) of an eval step, SURVEY.md section 8(d)."""
        def F(M, D, Do):
            return 2 * M * M + M * (2 * D + 5 * Do + 5)
        return self.S * (2 * F(self.M, self.D, self.dim_out_b) + F(self.M, self.D, self.dim_x))

    def bytes_per_state(self):
        """Compulsory HBM bytes per state of a fused eval step (f64), SURVEY.md section 8(d)."""
        return 8 * (self.dim_u + self.dim_y + self.S * (3 + self.dim_x + 2 * self.dim_out_b))


# BASELINE.json configs made concrete; S, dim_x, recog_len, k_factor and
# the noise/kernel initialisation come from the matching reference run script.
WORKLOADS = {
    # SpringNonlinear shape, small-scale settings (run/run_smallscale.py:31-52)
    'C1': Workload('C1-SpringNonlinear', dim_u=1, dim_y=1, dim_x=4, M=20, T=50, B=16, S=50, recog_len=16,
                   k_factor=50., loss_factors=(0.5, 0.), gp_len=2., learning_rate=0.1),
    # Actuator (run/run_smallscale.py:12,31-52)
    'C2': Workload('C2-Actuator', dim_u=1, dim_y=1, dim_x=4, M=50, T=100, B=64, S=50, recog_len=16,
                   k_factor=100., loss_factors=(0.5, 0.), gp_len=2., learning_rate=0.1),
    # Sarcos (run/run_sarcos.py:16-45)
    'C3': Workload('C3-Sarcos', dim_u=7, dim_y=7, dim_x=14, M=100, T=250, B=256, S=20, recog_len=16,
                   k_factor=50., loss_factors=(6., 0.), var_y=0.05 ** 2, learning_rate=0.05),
    # Sarcos, M=200, 256 sequences per GPU (global 2048 over 8 GPUs)
    'C4': Workload('C4-Sarcos-M200', dim_u=7, dim_y=7, dim_x=14, M=200, T=250, B=256, S=20, recog_len=16,
                   k_factor=50., loss_factors=(6., 0.), var_y=0.05 ** 2, learning_rate=0.05),
    # RoboMove synthetic (run/run_robomove.py:10-49 settings), 512 sequences per GPU (global 4096)
    'C5': Workload('C5-RoboMove-synthetic', dim_u=2, dim_y=2, dim_x=4, M=300, T=1000, B=512, S=50, recog_len=50,
                   k_factor=1., loss_factors=(10., 0.), zeta_mean=0.1 ** 2, var_x=0.1 ** 2, var_y=1.0,
                   gp_var=0.1 ** 2, learning_rate=0.01),
}


def tiny(name='tiny', **kw):
    """A seconds-scale workload for oracle/parity tests."""
    base = dict(dim_u=2, dim_y=2, dim_x=5, M=12, T=11, B=3, S=4, recog_len=3, k_factor=3.,
                loss_factors=(2., 0.7), gp_len=1.5, var_y=0.3 ** 2, var_x=0.05 ** 2)
    base.update(kw)
    return Workload(name, **base)


def softplus_inverse(y):
    """numpy inverse of softplus(x) + 1e-10 (cbfssm/model/tf_transform.py:13-16)."""
    y = np.asarray(y, dtype=np.float64)
    assert not np.any(y <= 1e-10), 'Input to backward transformation should be greater 1e-10'
    with np.errstate(over='ignore'):
        result = np.log(np.exp(y - 1e-10) - np.ones(1))
    return np.where(y > 35, y - 1e-10, result)


def make_params(w: Workload, seed=1):
    """Twelve unconstrained float64 parameter arrays at the run-script initialisation."""
    rng = np.random.default_rng(seed)
    p = {}
    for g, dout in (('f', w.dim_x), ('b', w.dim_out_b)):
        p[g + '.zeta_pos'] = rng.uniform(-w.zeta_pos, w.zeta_pos, size=(w.M, w.D))
        p[g + '.zeta_mean'] = w.zeta_mean * rng.random((w.M, dout))
        p[g + '.zeta_var_unc'] = softplus_inverse(w.zeta_var * np.ones((w.M, dout)))
        p[g + '.variance_unc'] = softplus_inverse(w.gp_var)               # shape (1,) like the reference
        p[g + '.lengthscales_unc'] = softplus_inverse(np.asarray([w.gp_len] * w.D))
    p['var_x_unc'] = softplus_inverse(np.asarray([w.var_x] * w.dim_x))
    p['var_y_unc'] = softplus_inverse(np.asarray([w.var_y] * w.dim_x))
    return {k: np.ascontiguousarray(v, dtype=np.float64) for k, v in p.items()}


def make_variant_params(w: Workload, variant='half', recog='rnn', seed=1):
    """(config, initial parameters) of the forward-only variants at run-script initial values: CBFSSMHALF
    (cbfssm/model/cbfssmhalf.py:20-47,82-93: gp_f only, var_y with dim_y entries, GRU(16) recognition model with TF's
    glorot-uniform / GRUCell initial values) or PRSSM (cbfssm/model/prssm.py:28-47: one shared lengthscale)."""
    cfg = w.model_config()
    cfg['var_y'] = np.asarray([w.var_y] * w.dim_y)
    cfg['recog_model'] = recog
    base = make_params(w, seed=seed)
    rng = np.random.default_rng(seed + 100)
    if variant == 'prssm':
        p = {k[2:]: v for k, v in base.items() if k.startswith('f.') and 'lengthscales' not in k}
        p['lengthscales_unc'] = softplus_inverse(np.asarray([w.gp_len]))
    else:
        p = {k: v for k, v in base.items() if k.startswith('f.')}
    p['var_x_unc'] = base['var_x_unc']
    p['var_y_unc'] = softplus_inverse(cfg['var_y'])

    def glorot(fan_in, fan_out):
        lim = np.sqrt(6.0 / (fan_in + fan_out))
        return rng.uniform(-lim, lim, size=(fan_in, fan_out))
    if recog == 'rnn':
        n_in, H = w.dim_u + w.dim_y, 16
        p.update({'recog.gate_kernel': glorot(n_in + H, 2 * H), 'recog.gate_bias': np.ones(2 * H),
                  'recog.cand_kernel': glorot(n_in + H, H), 'recog.cand_bias': np.zeros(H),
                  'recog.dense_kernel': glorot(H, w.dim_x), 'recog.dense_bias': np.zeros(w.dim_x)})
    return cfg, {k: np.ascontiguousarray(v, dtype=np.float64) for k, v in p.items()}


def perturb_params(p, seed=3, scale=0.2):
    """Move every parameter off its symmetric initial value (so per-dimension bugs cannot hide)."""
    rng = np.random.default_rng(seed)
    return {k: v + scale * rng.standard_normal(v.shape) * (0.2 if 'zeta_mean' in k else 1.0) for k, v in p.items()}


def softplus(x):
    """numpy softplus(x) + 1e-10 (cbfssm/model/tf_transform.py:19-21)."""
    return np.logaddexp(0.0, np.asarray(x, dtype=np.float64)) + 1e-10


def trained_like_params(w: Workload, ls_mult=4.0, zeta_mean=0.5, zeta_var=1e-2, cluster=1.0, seed=5):
    """Parameters of the kind training moves towards, for the conditioning sweep of the parity tests: lengthscales
    multiplied by `ls_mult` and inducing inputs contracted by `cluster` (correlated inducing points => K_mm
    ill-conditioned, the 1e-8 jitter of gp_tf.py:57,130 starts to matter), inducing means of order `zeta_mean` (the GP
    carries the dynamics), inducing variances log-uniform around `zeta_var` (one decade)."""
    pn = perturb_params(make_params(w, seed=1), scale=0.1)
    rng = np.random.default_rng(seed)
    for g in 'fb':
        pn[g + '.lengthscales_unc'] = softplus_inverse(softplus(pn[g + '.lengthscales_unc']) * ls_mult)
        pn[g + '.zeta_mean'] = zeta_mean * rng.standard_normal(pn[g + '.zeta_mean'].shape)
        pn[g + '.zeta_var_unc'] = softplus_inverse(zeta_var * np.exp(rng.uniform(-1.15, 1.15, pn[g + '.zeta_var_unc'].shape)))
        pn[g + '.zeta_pos'] = pn[g + '.zeta_pos'] * cluster
    return {k: np.ascontiguousarray(v, dtype=np.float64) for k, v in pn.items()}


def kmm_condition(p, g, jitter=1e-8):
    """2-norm condition number of K_mm + jitter I of GP `g` ('f' or 'b') at parameters p (host numpy; test helper)."""
    Z = p[g + '.zeta_pos'] / softplus(p[g + '.lengthscales_unc'])
    zs = np.sum(Z * Z, 1)
    K = softplus(p[g + '.variance_unc']) * np.exp(-0.5 * (-2.0 * Z @ Z.T + zs[:, None] + zs[None, :]))
    return float(np.linalg.cond(K + jitter * np.eye(K.shape[0])))


def make_inputs(w: Workload, seed=0):
    """u (B,T,dim_u), y (B,T,dim_y) ~ N(0,1): the reference z-normalises its data (datasets/base_ds.py:25-34)."""
    rng = np.random.default_rng(seed)
    u = rng.standard_normal((w.B, w.T, w.dim_u))
    y = rng.standard_normal((w.B, w.T, w.dim_y))
    return u, y


def make_noise(w: Workload, seed=2):
    rng = np.random.default_rng(seed)
    return {'hid_b': rng.standard_normal((2, w.T, w.B, w.S)),
            'eps_b': rng.standard_normal((2, w.T, w.B, w.S)),
            'eps_f': rng.standard_normal((w.T - 1, w.B, w.S))}


def workload_dict(w: Workload):
    return asdict(w)
