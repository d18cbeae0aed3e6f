"""CPU oracle for the CBF-SSM ELBO hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product path
(cbf-ssm_amd/) never does and fails loudly when its HIP library is missing.

PARITY UNPINNED: the reference (silvanmelchior/CBF-SSM) is pure Python on TensorFlow 1.8.0, TensorFlow is not
installable in the build container (no network, no cp310 wheel) and the reference ships no tests, golden vectors
or fixtures for this path.  This file is therefore a float64 numpy/scipy
restatement that follows the reference op for op; it is pinned only by independent cross-checks in
tests/test_oracle.py (scipy linear algebra, torch.distributions for the MVN KL / diag-normal log-prob,
closed-form known answers that follow from the reference code, finite differences for gradients).

Every function cites the reference file:line it restates (paths relative to /root/reference).
Noise and initial parameters are explicit inputs (the reference draws them unseeded inside the graph).
"""
import numpy as np
import scipy.linalg as sla

JITTER = 1e-8            # cbfssm/model/gp_tf.py:57 (cast_cholesky default)


# ---------------------------------------------------------------------------------------------------------------------
# cbfssm/model/tf_transform.py
# ---------------------------------------------------------------------------------------------------------------------
def tf_backward(y):
    """tf_transform.py:13-16 -- numpy inverse softplus with the y>35 branch and the positivity assert."""
    y = np.asarray(y, dtype=np.float64)
    assert not np.any(y <= 1e-10), 'Input to backward transformation should be greater 1e-10'
    with np.errstate(over='ignore'):
        result = np.log(np.exp(y - 1e-10) - np.ones(1))
    return np.where(y > 35, y - 1e-10, result)


def tf_forward(x):
    """tf_transform.py:19-21 -- tf.nn.softplus(x) + 1e-10 (softplus = log(1+exp(x)), evaluated stably)."""
    return np.logaddexp(0.0, np.asarray(x, dtype=np.float64)) + 1e-10


# ---------------------------------------------------------------------------------------------------------------------
# cbfssm/model/gp_tf.py
# ---------------------------------------------------------------------------------------------------------------------
class RBF:
    """gp_tf.py:20-49."""

    def __init__(self, variance_unc, lengthscales_unc):
        self.variance = tf_forward(variance_unc)            # gp_tf.py:27, shape (1,)
        self.lengthscales = tf_forward(lengthscales_unc)    # gp_tf.py:31

    def square_dist(self, X, X2):
        """gp_tf.py:33-43: -2 X X2^T + |X|^2 + |X2|^2 on lengthscale-scaled inputs, no clamp at zero."""
        X = X / self.lengthscales
        Xs = np.sum(np.square(X), 1)
        if X2 is None:
            return -2 * X @ X.T + Xs.reshape(-1, 1) + Xs.reshape(1, -1)
        X2 = X2 / self.lengthscales
        X2s = np.sum(np.square(X2), 1)
        return -2 * X @ X2.T + Xs.reshape(-1, 1) + X2s.reshape(1, -1)

    def Kdiag(self, X):
        """gp_tf.py:45-46."""
        return np.full((X.shape[0],), np.squeeze(self.variance))

    def K(self, X, X2=None):
        """gp_tf.py:48-49."""
        return self.variance * np.exp(-0.5 * self.square_dist(X, X2))


def cast_cholesky(mat, jitter=JITTER):
    """gp_tf.py:52-65: lower Cholesky of mat + jitter*I in float64.  Raises like TF when not PD."""
    mat = np.array(mat, dtype=np.float64, copy=True)
    mat[np.diag_indices_from(mat)] += jitter
    return np.linalg.cholesky(mat)



def conditional(Xnew, X, kern, f, q_sqrt, Lm=None):
    """gp_tf.py:68-100: the GPflow-1.0-style conditional with q_sqrt None, (M, Do) or (Do, M, M)."""
    num_func = f.shape[1]                                                              # :70
    Kmn = kern.K(X, Xnew)                                                              # :71
    if Lm is None:
        Lm = cast_cholesky(kern.K(X), jitter=1e-8)                                     # :72-73
    A = sla.solve_triangular(Lm, Kmn, lower=True)                                      # :76
    fvar = kern.Kdiag(Xnew) - np.sum(np.square(A), 0)                                  # :79
    fvar = np.tile(fvar[None, :], (num_func, 1))                                       # :80-81
    A = sla.solve_triangular(Lm.T, A, lower=False)                                     # :84
    fmean = A.T @ f                                                                    # :87
    if q_sqrt is not None:
        if q_sqrt.ndim == 2:
            LTA = A[None, :, :] * q_sqrt.T[:, :, None]                                 # :91
        elif q_sqrt.ndim == 3:
            A_tiled = np.tile(A[None, :, :], (num_func, 1, 1))                         # :93
            LTA = np.matmul(np.transpose(q_sqrt, (0, 2, 1)), A_tiled)                  # :94
        else:
            raise ValueError("bad dimension for q_sqrt")
        fvar = fvar + np.sum(np.square(LTA), 1)                                        # :98
    return fmean, fvar.T                                                               # :100

class GPModel:
    """gp_tf.py:103-172 with the trainable tensors passed in instead of drawn (gp_tf.py:112-123)."""

    def __init__(self, zeta_pos, zeta_mean, zeta_var_unc, variance_unc, lengthscales_unc):
        self.zeta_pos = np.asarray(zeta_pos, dtype=np.float64)
        self.zeta_mean = np.asarray(zeta_mean, dtype=np.float64)
        self.zeta_var = tf_forward(zeta_var_unc)                      # gp_tf.py:122
        self.zeta_std = np.sqrt(self.zeta_var)                        # gp_tf.py:123
        self.kern = RBF(variance_unc, lengthscales_unc)               # gp_tf.py:125-127
        self.num_points, self.in_dim = self.zeta_pos.shape
        self.out_dim = self.zeta_mean.shape[1]
        kernel_matrix = self.kern.K(self.zeta_pos)                    # gp_tf.py:129
        self.cholesky = cast_cholesky(kernel_matrix, jitter=JITTER)   # gp_tf.py:130

    def predict(self, Xnew):
        """gp_tf.py:132-161: sparse-GP conditional, diagonal q(z), unwhitened."""
        Kmn = self.kern.K(self.zeta_pos, Xnew)                                        # :134
        A = sla.solve_triangular(self.cholesky, Kmn, lower=True)                      # :137
        fvar = self.kern.Kdiag(Xnew) - np.sum(np.square(A), 0)                        # :140
        fvar = np.tile(fvar[None, :], (self.out_dim, 1))                              # :141-142
        A = sla.solve_triangular(self.cholesky.T, A, lower=False)                     # :145
        fmean = A.T @ self.zeta_mean                                                  # :148
        LTA = A[None, :, :] * self.zeta_std.T[:, :, None]                             # :152  (Do, M, N)
        fvar = fvar + np.sum(np.square(LTA), 1)                                       # :159
        return fmean, fvar.T                                                          # :161

    def prior_kl(self):
        """gp_tf.py:163-172: sum_d KL( N(mu_d, diag std_d^2) || N(0, L L^T) ), written out in closed form.

        tf.contrib.distributions.kl_divergence(MVNDiag, MVNTriL) evaluates
        0.5 * [ tr(K^-1 S) + mu^T K^-1 mu - M + log det K - log det S ]  per output dimension.
        (tests/test_oracle.py checks this against torch.distributions.)
        """
        L = self.cholesky
        M = self.num_points
        total = 0.0
        for d in range(self.out_dim):
            mu = self.zeta_mean[:, d]
            std = self.zeta_std[:, d]
            Linv_S = sla.solve_triangular(L, np.diag(std), lower=True)       # L^-1 S^{1/2}
            trace = np.sum(np.square(Linv_S))
            alpha = sla.solve_triangular(L, mu, lower=True)
            maha = np.sum(np.square(alpha))
            logdet_K = 2.0 * np.sum(np.log(np.diag(L)))
            logdet_S = 2.0 * np.sum(np.log(std))
            total += 0.5 * (trace + maha - M + logdet_K - logdet_S)
        return total


# ---------------------------------------------------------------------------------------------------------------------
# cbfssm/model/cbfssm.py
# ---------------------------------------------------------------------------------------------------------------------
def window_schedule(T, recog_len, run):
    """cbfssm.py:123-128: (resample, write) boolean arrays over t for backward run `run`."""
    t = np.arange(T)
    R = recog_len
    if run == 0:
        resample = np.mod(t + 1, 2 * R) == 0
        write = np.mod(t, 2 * R) < R
    else:
        resample = np.mod(t + R + 1, 2 * R) == 0
        write = np.mod(t, 2 * R) >= R
    return resample, write


class CBFSSMOracle:
    """cbfssm/model/cbfssm.py:10-271 as a function of explicit (params, u, y, noise, condition)."""

    def __init__(self, config, params):
        self.config = config
        self.dim_u = config['ds'].dim_u
        self.dim_y = config['ds'].dim_y
        self.dim_x = config['dim_x']
        p = params
        # cbfssm.py:30-48
        self.gp_f = GPModel(p['f.zeta_pos'], p['f.zeta_mean'], p['f.zeta_var_unc'],
                            p['f.variance_unc'], p['f.lengthscales_unc'])
        self.gp_b = GPModel(p['b.zeta_pos'], p['b.zeta_mean'], p['b.zeta_var_unc'],
                            p['b.variance_unc'], p['b.lengthscales_unc'])
        # cbfssm.py:51-54
        self.var_x = tf_forward(p['var_x_unc'])
        self.var_y = tf_forward(p['var_y_unc'])

    # -- cbfssm.py:101-158 ------------------------------------------------------------------------------------------
    def _backward_run(self, u, y, y2, prob, run, hid, eps, trace=None):
        B, T, _ = u.shape
        S = self.config['samples']
        dim_out = self.dim_x - self.dim_y
        R = self.config['recog_len']
        resample, write = window_schedule(T, R, run)
        h = np.zeros((B, S, dim_out))                                              # :106
        for t in range(T - 1, -1, -1):                                             # :107-111
            u_t = np.tile(u[:, t, None, :], (1, S, 1))                             # :74-76, :131
            y_t = np.tile(y[:, t, None, :], (1, S, 1))                             # :80-82, :132
            if resample[t]:
                hidden = np.tile(hid[t][:, :, None], (1, 1, dim_out))              # :133-135
            else:
                hidden = h                                                         # :136
            in_t = np.concatenate((hidden, u_t, y_t), axis=2)                      # :137
            fmean, fvar = self.gp_b.predict(in_t.reshape(B * S, self.dim_x + self.dim_u))   # :140-141
            fmean = fmean.reshape(B, S, dim_out) + in_t[:, :, :dim_out]            # :143,145
            fvar = fvar.reshape(B, S, dim_out) + self.var_x[:dim_out]              # :144,146
            e = np.tile(eps[t][:, :, None], (1, 1, dim_out))                       # :149
            out = fmean + e * np.sqrt(fvar)                                        # :150
            if write[t]:
                y2[t] = out                                                        # :151
                prob[t] = 0.5 * np.sum(np.log(2. * np.pi * np.e) + np.log(fvar))   # :154-156
            if trace is not None:
                trace.setdefault('b_h', {})[(run, t)] = out
                trace.setdefault('b_fmean', {})[(run, t)] = fmean
                trace.setdefault('b_fvar', {})[(run, t)] = fvar
            h = out                                                                # :158

    # -- cbfssm.py:84-99 --------------------------------------------------------------------------------------------
    def backward(self, u, y, noise, trace=None):
        B, T, _ = u.shape
        S = self.config['samples']
        dim_out = self.dim_x - self.dim_y
        y2 = np.zeros((T, B, S, dim_out))
        prob = np.zeros((T,))
        self._backward_run(u, y, y2, prob, 0, noise['hid_b'][0], noise['eps_b'][0], trace)   # :92
        self._backward_run(u, y, y2, prob, 1, noise['hid_b'][1], noise['eps_b'][1], trace)   # :93
        y2 = np.transpose(y2, (1, 0, 2, 3))                                         # :95
        out_dub = np.tile(y[:, :, None, :], (1, 1, S, 1))                           # :96
        y_tilde = np.concatenate((out_dub, y2), axis=3)                             # :97
        entropy = np.sum(prob)                                                      # :99
        return y_tilde, entropy

    # -- cbfssm.py:160-237 ------------------------------------------------------------------------------------------
    def forward(self, u, y_tilde, noise, condition, trace=None):
        B, T, _ = u.shape
        S = self.config['samples']
        R = self.config['recog_len']
        k_factor = self.config['k_factor']
        dim_x = self.dim_x
        x = np.zeros((T, B, S, dim_x))
        x[0] = y_tilde[:, 0]                                                        # :168-169
        prob = np.zeros((T - 1,))
        for t in range(T - 1):                                                      # :176-179
            u_t = np.tile(u[:, t, None, :], (1, S, 1))                              # :194
            x_t = x[t]                                                              # :195
            y_t = y_tilde[:, t + 1]                                                 # :196
            in_t = np.concatenate((x_t, u_t), axis=2)                               # :197
            fmean, fvar = self.gp_f.predict(in_t.reshape(B * S, self.dim_u + dim_x))   # :200-201
            fmean = fmean.reshape(B, S, dim_x) + in_t[:, :, :dim_x]                 # :203,205
            fvar = fvar.reshape(B, S, dim_x) + self.var_x                           # :204,206
            eps = np.tile(noise['eps_f'][t][:, :, None], (1, 1, dim_x))             # :209
            var_y_tiled = np.tile(self.var_y[None, None, :], (B, S, 1))             # :212-213
            var_y_tiled = var_y_tiled + (k_factor - 1.) * fvar                      # :214
            y_diff = y_t - fmean                                                    # :215
            s = var_y_tiled + fvar                                                  # :216
            k = fvar * (1.0 / s)                                                    # :217
            mu = fmean + k * y_diff                                                 # :218
            sig = np.ones((B, S, dim_x)) - k                                        # :219
            sig = np.square(sig) * fvar + np.square(k) * var_y_tiled                # :220
            x_cond = mu + eps * np.sqrt(sig)                                        # :221
            x_nocond = fmean + eps * np.sqrt(fvar)                                  # :224
            do_cond = bool(condition) or (t < R - 1)                                # :227
            x[t + 1] = x_cond if do_cond else x_nocond                              # :228-229
            kl_reg = np.log(fvar) - np.log(sig) + (sig + np.power(mu - fmean, 2.)) / fvar - 1.   # :232
            prob[t] = np.sum(kl_reg) * (0.5 if do_cond else 0.0)                    # :233-235
            if trace is not None:
                trace.setdefault('f_fmean', {})[t] = fmean
                trace.setdefault('f_fvar', {})[t] = fvar
        x_final = np.transpose(x, (1, 0, 2, 3))                                     # :181
        y_final = x_final[:, :, :, :self.dim_y]                                     # :182
        kl_x = np.sum(prob)                                                         # :183
        return x_final, y_final, kl_x

    # -- cbfssm.py:239-262 ------------------------------------------------------------------------------------------
    def loss_terms(self, y, y_final, kl_x, entropy):
        S = self.config['samples']
        lf = self.config['loss_factors']
        var = self.var_y[:self.dim_y]                                               # :245
        obs = y[:, :, None, :]                                                      # :249
        # MultivariateNormalDiag(loc, scale_diag=sqrt(var)).log_prob(obs)           # :247-250
        log_probs = -0.5 * np.sum(np.square(obs - y_final) / var + np.log(2 * np.pi) + np.log(var), axis=-1)
        loglik = np.sum(log_probs)                                                  # :251
        kl_z_f = self.gp_f.prior_kl()                                               # :254
        kl_z_b = self.gp_b.prior_kl()                                               # :255
        divisor = 1.0 / S                                                           # :257
        elbo = (loglik * lf[0] * divisor - kl_x * lf[0] * divisor
                + entropy * lf[1] * divisor - kl_z_f - kl_z_b)                      # :258-261
        return {'loglik': loglik, 'kl_x': kl_x, 'entropy': entropy, 'kl_z_f': kl_z_f, 'kl_z_b': kl_z_b,
                'elbo': elbo, 'loss': -elbo}                                        # :262

    # -- cbfssm.py:264-271 ------------------------------------------------------------------------------------------
    def prediction(self, y, x_final, y_final):
        pred_mean = np.mean(y_final, axis=2)                                        # :267 (tf.nn.moments: population)
        pred_var = np.var(y_final, axis=2) + self.var_y[:self.dim_y]                # :267-268
        internal_mean = np.mean(x_final, axis=2)                                    # :269
        internal_var = np.var(x_final, axis=2)
        mse = np.mean(np.square(y - pred_mean))                                     # :270
        sde = np.abs(pred_mean - y) / np.sqrt(pred_var)                             # :271
        return {'pred_mean': pred_mean, 'pred_var': pred_var, 'internal_mean': internal_mean,
                'internal_var': internal_var, 'mse': mse, 'sde': sde}

    def run(self, u, y, noise, condition=True, trace=None):
        """One execution of the graph on one mini-batch: everything sess.run could fetch."""
        u = np.asarray(u, dtype=np.float64)
        y = np.asarray(y, dtype=np.float64)
        y_tilde, entropy = self.backward(u, y, noise, trace)
        x_final, y_final, kl_x = self.forward(u, y_tilde, noise, condition, trace)
        out = self.loss_terms(y, y_final, kl_x, entropy)
        out.update(self.prediction(y, x_final, y_final))
        out.update({'y_tilde': y_tilde, 'x_final': x_final})
        return out


def elbo_step(config, params, u, y, noise, condition=True, trace=None):
    return CBFSSMOracle(config, params).run(u, y, noise, condition, trace)


# ---------------------------------------------------------------------------------------------------------------------
# cbfssm/model/cbfssmhalf.py  (forward-only variant: recognition model for x_0, Kalman update on the observed dims)
# ---------------------------------------------------------------------------------------------------------------------
def gru_recognition(recog, u, y, recog_len):
    """cbfssmhalf.py:64-95, recog == 'rnn': TF-1.8 GRUCell(16) over the reversed first recog_len steps of [u, y],
    then a dense layer to dim_x.  recog: dict gate_kernel (in+16, 32), gate_bias (32), cand_kernel (in+16, 16),
    cand_bias (16), dense_kernel (16, dim_x), dense_bias (dim_x)."""
    uy = np.concatenate((u, y), axis=2)[:, :recog_len, :][:, ::-1, :]             # :77-78,83
    h = np.zeros((u.shape[0], recog['cand_bias'].shape[0]))                       # :80

    def sigmoid(v):
        return 1.0 / (1.0 + np.exp(-v))
    for t in range(uy.shape[1]):
        x = uy[:, t, :]
        gates = sigmoid(np.concatenate((x, h), 1) @ recog['gate_kernel'] + recog['gate_bias'])
        r, z = np.split(gates, 2, axis=1)
        c = np.tanh(np.concatenate((x, r * h), 1) @ recog['cand_kernel'] + recog['cand_bias'])
        h = z * h + (1.0 - z) * c
    return h @ recog['dense_kernel'] + recog['dense_bias']                        # :86


class CBFSSMHALFOracle:
    """cbfssm/model/cbfssmhalf.py:7-211 as a function of explicit (params, u, y, eps_f, condition)."""

    def __init__(self, config, params):
        self.config = config
        self.dim_u, self.dim_y, self.dim_x = config['ds'].dim_u, config['ds'].dim_y, config['dim_x']
        p = params
        self.gp_f = GPModel(p['f.zeta_pos'], p['f.zeta_mean'], p['f.zeta_var_unc'],
                            p['f.variance_unc'], p['f.lengthscales_unc'])          # :24-33
        self.var_x = tf_forward(p['var_x_unc'])                                      # :36-37
        self.var_y = tf_forward(p['var_y_unc'])                                      # :38-39  (dim_y entries)
        self.params = params

    def recog(self, u, y):
        kind = self.config.get('recog_model', 'rnn')                                 # :71-74
        if kind == 'output':                                                         # :76-80
            x0 = np.concatenate((y[:, 0, :], np.zeros((y.shape[0], self.dim_x - self.dim_y))), axis=1)
        else:
            x0 = gru_recognition({k[6:]: v for k, v in self.params.items() if k.startswith('recog.')}, u, y,
                                 self.config['recog_len'])
        return x0

    def run(self, u, y, noise, condition=True):
        u, y = np.asarray(u, dtype=np.float64), np.asarray(y, dtype=np.float64)
        B, T, _ = u.shape
        S, R, kf = self.config['samples'], self.config['recog_len'], self.config['k_factor']
        dim_x, dim_y = self.dim_x, self.dim_y
        x = np.zeros((T, B, S, dim_x))
        x[0] = np.tile(self.recog(u, y)[:, None, :], (1, S, 1))                      # :80,87,106-107
        prob = np.zeros((T - 1,))
        for t in range(T - 1):
            u_t = np.tile(u[:, t, None, :], (1, S, 1))
            y_t = np.tile(y[:, t + 1, None, :], (1, S, 1))                           # :133
            in_t = np.concatenate((x[t], u_t), axis=2)
            fmean, fvar = self.gp_f.predict(in_t.reshape(B * S, self.dim_u + dim_x))
            fmean = fmean.reshape(B, S, dim_x) + in_t[:, :, :dim_x]                  # :140-142
            fvar = fvar.reshape(B, S, dim_x) + self.var_x                            # :143
            eps = np.tile(noise['eps_f'][t][:, :, None], (1, 1, dim_x))              # :146
            var_y_t = np.tile(self.var_y[None, None, :], (B, S, 1)) + (kf - 1.) * fvar[:, :, :dim_y]   # :149-151
            y_diff = y_t - fmean[:, :, :dim_y]                                       # :152
            s = var_y_t + fvar[:, :, :dim_y]                                         # :153
            k = fvar[:, :, :dim_y] * (1.0 / s)                                       # :154
            pad = np.zeros((B, S, dim_x - dim_y))
            mu = fmean + np.concatenate((k * y_diff, pad), axis=2)                   # :156
            sig = np.square(np.ones((B, S, dim_x)) - np.concatenate((k, pad), axis=2)) * fvar   # :157-158
            sig = sig + np.concatenate((np.square(k) * var_y_t, pad), axis=2)        # :159
            x_cond = mu + eps * np.sqrt(sig)                                         # :160
            x_nocond = fmean + eps * np.sqrt(fvar)                                   # :163
            do_cond = bool(condition) or (t < R - 1)                                 # :166
            x[t + 1] = x_cond if do_cond else x_nocond
            kl_reg = np.log(fvar) - np.log(sig) + (sig + np.power(mu - fmean, 2.)) / fvar - 1.   # :171
            prob[t] = np.sum(kl_reg) * (0.5 if do_cond else 0.0)
        x_final = np.transpose(x, (1, 0, 2, 3))
        y_final = x_final[..., :dim_y]
        kl_x = np.sum(prob)
        var = self.var_y[:dim_y]
        log_probs = -0.5 * np.sum(np.square(y[:, :, None, :] - y_final) / var + np.log(2 * np.pi) + np.log(var), axis=-1)
        loglik = np.sum(log_probs)                                                   # :181-189
        kl_z_f = self.gp_f.prior_kl()                                                # :192
        lf = self.config['loss_factors']
        elbo = loglik * lf[0] / S - kl_x * lf[0] / S - kl_z_f                        # :195-198
        return {'loss': -elbo, 'loglik': loglik, 'kl_x': kl_x, 'kl_z_f': kl_z_f, 'x_final': x_final,
                'pred_mean': np.mean(y_final, axis=2), 'pred_var': np.var(y_final, axis=2) + var}
