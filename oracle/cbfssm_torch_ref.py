"""PyTorch-CPU float64 restatement of the CBF-SSM ELBO step, op for op.  TEST INFRASTRUCTURE ONLY.

Same role and the same restrictions as oracle/cbfssm_oracle.py (PARITY UNPINNED, see its header): only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.  It exists for two things the numpy oracle
cannot give:
  * gradients of the loss w.r.t. the twelve trainable tensors by reverse-mode autodiff (what
    tf.train.AdamOptimizer.minimize differentiates, cbfssm/model/cbfssm.py:273-275) -- checked against central
    finite differences of the numpy oracle in tests/test_oracle.py;
  * the CPU baseline: the same *unfused* op sequence TensorFlow 1.8 executes on Eigen/MKL kernels (explicit
    -2XX'+|x|^2+|x'|^2 RBF, Cholesky, two triangular solves, materialised (Do,M,N) variance term, Python time
    loop), timed on the host cores next to the GPU number.

Every block cites the reference file:line it restates (paths relative to /root/reference).
"""
import math
import numpy as np
import torch

JITTER = 1e-8


def tf_forward(x):
    """cbfssm/model/tf_transform.py:19-21."""
    return torch.nn.functional.softplus(x, beta=1.0, threshold=1e9) + 1e-10


class RBF:
    """cbfssm/model/gp_tf.py:20-49."""

    def __init__(self, variance_unc, lengthscales_unc):
        self.variance = tf_forward(variance_unc)
        self.lengthscales = tf_forward(lengthscales_unc)

    def square_dist(self, X, X2):
        X = X / self.lengthscales
        Xs = torch.sum(torch.square(X), 1)
        if X2 is None:
            return -2 * X @ X.T + Xs.reshape(-1, 1) + Xs.reshape(1, -1)
        X2 = X2 / self.lengthscales
        X2s = torch.sum(torch.square(X2), 1)
        return -2 * X @ X2.T + Xs.reshape(-1, 1) + X2s.reshape(1, -1)

    def K(self, X, X2=None):
        return self.variance * torch.exp(-0.5 * self.square_dist(X, X2))


class GPModel:
    """cbfssm/model/gp_tf.py:103-172."""

    def __init__(self, zeta_pos, zeta_mean, zeta_var_unc, variance_unc, lengthscales_unc):
        self.zeta_pos = zeta_pos
        self.zeta_mean = zeta_mean
        self.zeta_var = tf_forward(zeta_var_unc)
        self.zeta_std = torch.sqrt(self.zeta_var)
        self.kern = RBF(variance_unc, lengthscales_unc)
        self.num_points = zeta_pos.shape[0]
        self.out_dim = zeta_mean.shape[1]
        Kmm = self.kern.K(zeta_pos)                                                      # :129
        # cast_cholesky (gp_tf.py:57-65): the factorisation runs in float64 whatever the model dtype, then casts back
        Kmm64 = Kmm.to(torch.float64)                                                    # :59-60
        Kmm64 = Kmm64 + JITTER * torch.eye(self.num_points, dtype=torch.float64)         # :52-54
        self.cholesky = torch.linalg.cholesky(Kmm64).to(Kmm.dtype)                       # :130, :62-64

    def predict(self, Xnew):
        Kmn = self.kern.K(self.zeta_pos, Xnew)                                           # :134
        A = torch.linalg.solve_triangular(self.cholesky, Kmn, upper=False)               # :137
        fvar = torch.squeeze(self.kern.variance) - torch.sum(torch.square(A), 0)         # :140
        fvar = fvar[None, :].repeat(self.out_dim, 1)                                     # :141-142
        A = torch.linalg.solve_triangular(self.cholesky.T, A, upper=True)                # :145
        fmean = A.T @ self.zeta_mean                                                     # :148
        LTA = A[None, :, :] * self.zeta_std.T[:, :, None]                                # :152
        fvar = fvar + torch.sum(torch.square(LTA), 1)                                    # :159
        return fmean, fvar.T                                                             # :161

    def prior_kl(self):
        """gp_tf.py:163-172 through torch.distributions (independent of the numpy oracle's closed form)."""
        D = torch.distributions
        prior = D.MultivariateNormal(torch.zeros(self.out_dim, self.num_points, dtype=self.cholesky.dtype),
                                     scale_tril=self.cholesky[None].repeat(self.out_dim, 1, 1))
        post = D.MultivariateNormal(self.zeta_mean.T, scale_tril=torch.diag_embed(self.zeta_std.T))
        return torch.sum(D.kl_divergence(post, prior))


def window_flags(t, R, run):
    """cbfssm/model/cbfssm.py:123-128."""
    if run == 0:
        return (t + 1) % (2 * R) == 0, t % (2 * R) < R
    return (t + R + 1) % (2 * R) == 0, t % (2 * R) >= R


def elbo_step(config, params, u, y, noise, condition=True, want_pred=False):
    """cbfssm/model/cbfssm.py:25-271.  params / u / y / noise: float64 torch tensors (params may require grad)."""
    dim_u, dim_y, dim_x = config['ds'].dim_u, config['ds'].dim_y, config['dim_x']
    S, R, k_factor = config['samples'], config['recog_len'], config['k_factor']
    lf = config['loss_factors']
    dim_out = dim_x - dim_y
    B, T, _ = u.shape
    p = params
    gp_f = GPModel(p['f.zeta_pos'], p['f.zeta_mean'], p['f.zeta_var_unc'], p['f.variance_unc'], p['f.lengthscales_unc'])
    gp_b = GPModel(p['b.zeta_pos'], p['b.zeta_mean'], p['b.zeta_var_unc'], p['b.variance_unc'], p['b.lengthscales_unc'])
    var_x = tf_forward(p['var_x_unc'])                                                   # :51-54
    var_y = tf_forward(p['var_y_unc'])

    u_dub = u.permute(1, 0, 2)[:, :, None, :].repeat(1, 1, S, 1)                         # :74-76 (physical tile)
    y_dub = y.permute(1, 0, 2)[:, :, None, :].repeat(1, 1, S, 1)                         # :80-82

    # ---- backward, two runs (cbfssm.py:84-158)
    y2 = [None] * T
    prob = [None] * T
    c = math.log(2. * math.pi * math.e)
    for run in (0, 1):
        h = torch.zeros(B, S, dim_out, dtype=u.dtype)                                    # :106
        for t in range(T - 1, -1, -1):
            resample, write = window_flags(t, R, run)
            hidden = noise['hid_b'][run, t][:, :, None].repeat(1, 1, dim_out) if resample else h   # :133-136
            in_t = torch.cat((hidden, u_dub[t], y_dub[t]), dim=2)                        # :137
            fmean, fvar = gp_b.predict(in_t.reshape(B * S, dim_x + dim_u))               # :140-141
            fmean = fmean.reshape(B, S, dim_out) + in_t[:, :, :dim_out]                  # :143-145
            fvar = fvar.reshape(B, S, dim_out) + var_x[:dim_out]                         # :144-146
            eps = noise['eps_b'][run, t][:, :, None].repeat(1, 1, dim_out)               # :149
            out = fmean + eps * torch.sqrt(fvar)                                         # :150
            if write:
                y2[t] = out                                                              # :151
                prob[t] = 0.5 * torch.sum(c + torch.log(fvar))                           # :154-156
            h = out                                                                      # :158
    y2 = torch.stack(y2).permute(1, 0, 2, 3)                                             # :95
    y_tilde = torch.cat((y[:, :, None, :].repeat(1, 1, S, 1), y2), dim=3)                # :96-97
    entropy = torch.sum(torch.stack(prob))                                               # :99

    # ---- forward (cbfssm.py:160-237)
    xs = [y_tilde[:, 0]]                                                                 # :168-169
    probf = []
    y_tilde_t = y_tilde.permute(1, 0, 2, 3)                                              # :173
    for t in range(T - 1):
        x_t = xs[t]
        in_t = torch.cat((x_t, u_dub[t]), dim=2)                                         # :197
        fmean, fvar = gp_f.predict(in_t.reshape(B * S, dim_u + dim_x))                   # :200-201
        fmean = fmean.reshape(B, S, dim_x) + in_t[:, :, :dim_x]                          # :203-205
        fvar = fvar.reshape(B, S, dim_x) + var_x                                         # :204-206
        eps = noise['eps_f'][t][:, :, None].repeat(1, 1, dim_x)                          # :209
        var_y_tiled = var_y[None, None, :].repeat(B, S, 1) + (k_factor - 1.) * fvar      # :212-214
        y_diff = y_tilde_t[t + 1] - fmean                                                # :215
        s = var_y_tiled + fvar                                                           # :216
        k = fvar * torch.reciprocal(s)                                                   # :217
        mu = fmean + k * y_diff                                                          # :218
        sig = torch.square(1.0 - k) * fvar + torch.square(k) * var_y_tiled               # :219-220
        x_cond = mu + eps * torch.sqrt(sig)                                              # :221
        x_nocond = fmean + eps * torch.sqrt(fvar)                                        # :224
        do_cond = bool(condition) or (t < R - 1)                                         # :227
        xs.append(x_cond if do_cond else x_nocond)                                       # :228-229
        kl_reg = torch.log(fvar) - torch.log(sig) + (sig + torch.pow(mu - fmean, 2.)) / fvar - 1.   # :232
        probf.append(torch.sum(kl_reg) * (0.5 if do_cond else 0.0))                      # :233-235
    x_final = torch.stack(xs).permute(1, 0, 2, 3)                                        # :181
    y_final = x_final[:, :, :, :dim_y]                                                   # :182
    kl_x = torch.sum(torch.stack(probf)) if probf else torch.zeros((), dtype=u.dtype)    # :183

    # ---- loss (cbfssm.py:239-262)
    var_full = var_y[:dim_y][None, None, None, :].repeat(B, T, S, 1)                     # :245-246
    obs = y[:, :, None, :].repeat(1, 1, S, 1)                                            # :249
    y_dist = torch.distributions.Independent(torch.distributions.Normal(y_final, torch.sqrt(var_full)), 1)   # :247
    loglik = torch.sum(y_dist.log_prob(obs))                                             # :250-251
    kl_z_f = gp_f.prior_kl()                                                             # :254
    kl_z_b = gp_b.prior_kl()                                                             # :255
    divisor = 1.0 / S
    elbo = (loglik * float(lf[0]) * divisor - kl_x * float(lf[0]) * divisor
            + entropy * float(lf[1]) * divisor - kl_z_f - kl_z_b)                        # :258-261
    out = {'loss': -elbo, 'loglik': loglik, 'kl_x': kl_x, 'entropy': entropy, 'kl_z_f': kl_z_f, 'kl_z_b': kl_z_b}
    if want_pred:
        out['pred_mean'] = torch.mean(y_final, dim=2)                                    # :267
        out['pred_var'] = torch.var(y_final, dim=2, unbiased=False) + var_y[:dim_y]      # :267-268
        out['x_final'] = x_final
        out['y_tilde'] = y_tilde
    return out


def loss_and_grads(config, params_np, u_np, y_np, noise_np, condition=True, dtype=torch.float64):
    """Loss and d loss / d(12 unconstrained tensors) by reverse-mode autodiff on the CPU; dtype = torch.float32 is the
    reference's float32 model (cbfssm.py:12: every tensor float32, the Cholesky through float64, gp_tf.py:57-65)."""
    params = {k: torch.tensor(v, dtype=dtype, requires_grad=True) for k, v in params_np.items()}
    u = torch.tensor(u_np, dtype=dtype)
    y = torch.tensor(y_np, dtype=dtype)
    noise = {k: torch.tensor(v, dtype=dtype) for k, v in noise_np.items()}
    out = elbo_step(config, params, u, y, noise, condition)
    out['loss'].backward()
    grads = {k: v.grad.detach().numpy().copy() for k, v in params.items()}
    return {k: float(v.detach()) for k, v in out.items()}, grads


# ---------------------------------------------------------------------------------------------------------------------
# cbfssm/model/cbfssmhalf.py
# ---------------------------------------------------------------------------------------------------------------------
def gru_recognition(recog, u, y, recog_len):
    """cbfssmhalf.py:82-93: TF-1.8 GRUCell(16) on the reversed first recog_len steps of [u, y] + dense to dim_x."""
    uy = torch.flip(torch.cat((u, y), dim=2)[:, :recog_len, :], dims=[1])
    h = torch.zeros(u.shape[0], recog['cand_bias'].shape[0], dtype=u.dtype, device=u.device)
    for t in range(uy.shape[1]):
        x = uy[:, t, :]
        gates = torch.sigmoid(torch.cat((x, h), 1) @ recog['gate_kernel'] + recog['gate_bias'])
        r, z = torch.chunk(gates, 2, dim=1)
        c = torch.tanh(torch.cat((x, r * h), 1) @ recog['cand_kernel'] + recog['cand_bias'])
        h = z * h + (1.0 - z) * c
    return h @ recog['dense_kernel'] + recog['dense_bias']


def half_elbo_step(config, params, u, y, noise, condition=True):
    """cbfssm/model/cbfssmhalf.py:20-199 on float64 CPU tensors (params may require grad)."""
    dim_u, dim_y, dim_x = config['ds'].dim_u, config['ds'].dim_y, config['dim_x']
    S, R, kf = config['samples'], config['recog_len'], config['k_factor']
    lf = config['loss_factors']
    B, T, _ = u.shape
    p = params
    gp_f = GPModel(p['f.zeta_pos'], p['f.zeta_mean'], p['f.zeta_var_unc'], p['f.variance_unc'], p['f.lengthscales_unc'])
    var_x, var_y = tf_forward(p['var_x_unc']), tf_forward(p['var_y_unc'])
    if config.get('recog_model', 'rnn') == 'output':
        x0 = torch.cat((y[:, 0, :], torch.zeros(B, dim_x - dim_y, dtype=u.dtype)), dim=1)
    else:
        x0 = gru_recognition({k[6:]: v for k, v in p.items() if k.startswith('recog.')}, u, y, R)
    xs = [x0[:, None, :].repeat(1, S, 1)]
    probf = []
    for t in range(T - 1):
        u_t = u[:, t, None, :].repeat(1, S, 1)
        y_t = y[:, t + 1, None, :].repeat(1, S, 1)
        in_t = torch.cat((xs[t], u_t), dim=2)
        fmean, fvar = gp_f.predict(in_t.reshape(B * S, dim_u + dim_x))
        fmean = fmean.reshape(B, S, dim_x) + in_t[:, :, :dim_x]
        fvar = fvar.reshape(B, S, dim_x) + var_x
        eps = noise['eps_f'][t][:, :, None].repeat(1, 1, dim_x)
        var_y_t = var_y[None, None, :].repeat(B, S, 1) + (kf - 1.) * fvar[:, :, :dim_y]
        y_diff = y_t - fmean[:, :, :dim_y]
        s = var_y_t + fvar[:, :, :dim_y]
        k = fvar[:, :, :dim_y] * torch.reciprocal(s)
        pad = torch.zeros(B, S, dim_x - dim_y, dtype=u.dtype)
        mu = fmean + torch.cat((k * y_diff, pad), dim=2)
        sig = torch.square(1.0 - torch.cat((k, pad), dim=2)) * fvar + torch.cat((torch.square(k) * var_y_t, pad), dim=2)
        x_cond = mu + eps * torch.sqrt(sig)
        x_nocond = fmean + eps * torch.sqrt(fvar)
        do_cond = bool(condition) or (t < R - 1)
        xs.append(x_cond if do_cond else x_nocond)
        kl_reg = torch.log(fvar) - torch.log(sig) + (sig + torch.pow(mu - fmean, 2.)) / fvar - 1.
        probf.append(torch.sum(kl_reg) * (0.5 if do_cond else 0.0))
    x_final = torch.stack(xs).permute(1, 0, 2, 3)
    y_final = x_final[..., :dim_y]
    kl_x = torch.sum(torch.stack(probf)) if probf else torch.zeros((), dtype=u.dtype)
    var = var_y[:dim_y]
    loglik = torch.sum(torch.distributions.Normal(y_final, torch.sqrt(var)).log_prob(y[:, :, None, :]))
    kl_z_f = gp_f.prior_kl()
    elbo = loglik * float(lf[0]) / S - kl_x * float(lf[0]) / S - kl_z_f
    return {'loss': -elbo, 'loglik': loglik, 'kl_x': kl_x, 'kl_z_f': kl_z_f, 'x_final': x_final}


def half_loss_and_grads(config, params_np, u_np, y_np, noise_np, condition=True):
    params = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in params_np.items()}
    noise = {k: torch.tensor(v, dtype=torch.float64) for k, v in noise_np.items()}
    out = half_elbo_step(config, params, torch.tensor(u_np), torch.tensor(y_np), noise, condition)
    out['loss'].backward()
    return float(out['loss'].detach()), {k: (v.grad.detach().numpy().copy() if v.grad is not None else np.zeros(v.shape))
                                for k, v in params.items()}


# ---------------------------------------------------------------------------------------------------------------------
# cbfssm/model/prssm.py  (PR-SSM baseline: free-running forward pass, ELBO = lambda0 * loglik - KL_z)
# ---------------------------------------------------------------------------------------------------------------------
def conv_recognition(recog, u, y, recog_len):
    """prssm.py:143-155: conv1d(5 filters, width 3, relu) -> max_pool(2,2) -> dense, computed in float32."""
    uy = torch.cat((u, y), dim=2)[:, :recog_len, :].to(torch.float32)                     # (B, L, C)
    k = recog['conv_kernel'].to(torch.float32)                                            # (3, C, 5) TF layout
    x = torch.nn.functional.conv1d(uy.permute(0, 2, 1), k.permute(2, 1, 0), recog['conv_bias'].to(torch.float32))
    x = torch.relu(x)                                                                     # (B, 5, L-2)
    x = torch.nn.functional.max_pool1d(x, 2, 2)                                           # (B, 5, (L-2)//2)
    x = x.permute(0, 2, 1).reshape(u.shape[0], -1)                                        # TF flattens (time, channel)
    out = x @ recog['dense_kernel'].to(torch.float32) + recog['dense_bias'].to(torch.float32)
    return out.to(torch.float64)


def prssm_elbo_step(config, params, u, y, noise):
    """cbfssm/model/prssm.py:19-130 on float64 CPU tensors."""
    dim_u, dim_y, dim_x = config['ds'].dim_u, config['ds'].dim_y, config['dim_x']
    S = config['samples']
    lf = config['loss_factors']
    B, T, _ = u.shape
    p = params
    M = p['zeta_pos'].shape[0]
    zeta_var = tf_forward(p['zeta_var_unc'])
    var_x, var_y = tf_forward(p['var_x_unc']), tf_forward(p['var_y_unc'])
    kern = RBF(p['variance_unc'], p['lengthscales_unc'])                                  # scalar lengthscale, :40
    recog = config['recog_model']
    rp = {k[6:]: v for k, v in p.items() if k.startswith('recog.')}
    if recog == 'output':
        x0 = torch.cat((y[:, 0, :], torch.zeros(B, dim_x - dim_y, dtype=u.dtype)), dim=1)
    elif recog == 'conv':
        x0 = conv_recognition(rp, u, y, config['recog_len'])
    else:
        x0 = gru_recognition(rp, u, y, config['recog_len'])
    # conditional() with Lm=None: jittered Cholesky per call (gp_tf.py:68-100) -- loop invariant
    Lm = torch.linalg.cholesky(kern.K(p['zeta_pos']) + JITTER * torch.eye(M, dtype=u.dtype))
    q_sqrt = torch.sqrt(zeta_var)
    xs = [x0[:, None, :].repeat(1, S, 1)]
    for t in range(T - 1):
        u_t = u[:, t, None, :].repeat(1, S, 1)
        in_t = torch.cat((xs[t], u_t), dim=2).reshape(B * S, dim_u + dim_x)
        Kmn = kern.K(p['zeta_pos'], in_t)
        A = torch.linalg.solve_triangular(Lm, Kmn, upper=False)
        fvar = torch.squeeze(kern.variance) - torch.sum(torch.square(A), 0)
        fvar = fvar[None, :].repeat(dim_x, 1)
        A = torch.linalg.solve_triangular(Lm.T, A, upper=True)
        fmean = A.T @ p['zeta_mean']
        LTA = A[None, :, :] * q_sqrt.T[:, :, None]
        fvar = (fvar + torch.sum(torch.square(LTA), 1)).T
        fmean = fmean.reshape(B, S, dim_x) + xs[t]
        fvar = fvar.reshape(B, S, dim_x) + var_x
        eps = noise['eps_f'][t][:, :, None].repeat(1, 1, dim_x)
        xs.append(fmean + eps * torch.sqrt(fvar))                                          # prssm.py:122-125
    x_final = torch.stack(xs).permute(1, 0, 2, 3)
    y_final = x_final[..., :dim_y]
    loglik = torch.sum(torch.distributions.Normal(y_final, torch.sqrt(var_y)).log_prob(y[:, :, None, :]))
    # KL regulariser: prior Cholesky WITHOUT jitter (prssm.py:81-82)
    D = torch.distributions
    prior = D.MultivariateNormal(torch.zeros(dim_x, M, dtype=u.dtype),
                                 scale_tril=torch.linalg.cholesky(kern.K(p['zeta_pos']))[None].repeat(dim_x, 1, 1))
    post = D.MultivariateNormal(p['zeta_mean'].T, scale_tril=torch.diag_embed(torch.sqrt(zeta_var).T))
    kl_reg = torch.sum(D.kl_divergence(post, prior))
    elbo = loglik * float(lf[0]) - kl_reg                                                  # prssm.py:96
    return {'loss': -elbo, 'loglik': loglik, 'kl_z': kl_reg, 'x_final': x_final,
            'pred_mean': torch.mean(y_final, dim=2), 'pred_var': torch.var(y_final, dim=2, unbiased=False) + var_y}


def prssm_loss_and_grads(config, params_np, u_np, y_np, noise_np):
    params = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in params_np.items()}
    noise = {k: torch.tensor(v, dtype=torch.float64) for k, v in noise_np.items()}
    out = prssm_elbo_step(config, params, torch.tensor(u_np), torch.tensor(y_np), noise)
    out['loss'].backward()
    res = {k: v.detach().numpy() for k, v in out.items()}
    return res, {k: (v.grad.detach().numpy().copy() if v.grad is not None else np.zeros(v.shape))
                 for k, v in params.items()}
