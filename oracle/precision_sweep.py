"""BASELINE.json configs[4] asks for an fp32-vs-bf16 ELBO tolerance sweep of the RoboMove-shaped problem (M = 300).
This is the CPU emulation of that sweep on the oracle (test infrastructure, not a product path): the GP conditional
(reference cbfssm/model/gp_tf.py:132-161) is evaluated with reduced-precision operands while the recurrence around
it stays float64, and the ELBO / predictive moments are compared with the all-float64 evaluation.

    python oracle/precision_sweep.py            # prints the table quoted in DESIGN.md section 6

Variants of GPModel.predict:
  trsm-fp32         the reference's own fp32 mode: Cholesky computed in f64 and cast (gp_tf.py:57-65), K_mn and the two
                    triangular solves in fp32
  contraction-fp32  K^-1 (from the f64 Cholesky) cast to fp32, A2 = K^-1 k as one fp32 product (what an fp32 MFMA kernel
                    of this build's formulation would compute)
  contraction-bf16  the same with K^-1 and k rounded to bf16, fp32 accumulation (a bf16 MFMA kernel)
"""
import os
import sys
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'cbf-ssm_amd')]
from cbfssm import synthetic as syn            # noqa: E402
from oracle import cbfssm_torch_ref as tref    # noqa: E402

F64 = tref.GPModel.predict


def make_predict(kind):
    def predict(self, Xnew):
        if kind == 'f64':
            return F64(self, Xnew)
        dt = torch.float32
        var = torch.squeeze(self.kern.variance).to(dt)
        Kmn = self.kern.K(self.zeta_pos.to(dt), Xnew.to(dt)) if False else self.kern.K(self.zeta_pos, Xnew).to(dt)
        mu, s2 = self.zeta_mean.to(dt), self.zeta_var.to(dt)
        if kind == 'trsm-fp32':
            L = self.cholesky.to(dt)
            A = torch.linalg.solve_triangular(L, Kmn, upper=False)
            fvar0 = var - torch.sum(A * A, 0)
            A2 = torch.linalg.solve_triangular(L.T, A, upper=True)
        else:
            Kinv = torch.cholesky_inverse(self.cholesky)
            if kind == 'contraction-bf16':
                Kinv_r, k_r = Kinv.to(torch.bfloat16).to(dt), Kmn.to(torch.bfloat16).to(dt)
            else:
                Kinv_r, k_r = Kinv.to(dt), Kmn
            A2 = Kinv_r @ k_r
            fvar0 = var - torch.sum(k_r * A2, 0)
        fmean = A2.T @ mu
        fvar = fvar0[:, None] + (A2 * A2).T @ s2
        return fmean.to(torch.float64), fvar.to(torch.float64)
    return predict


def run(w, tag, trained_like=False):
    cfg = w.model_config()
    pn = syn.perturb_params(syn.make_params(w, seed=1), scale=0.1)
    if trained_like:
        # what training moves towards: correlated inducing points (lengthscale x4 => K_mm ill-conditioned, the 1e-8
        # jitter starts to matter) and inducing means of order one (the GP carries the dynamics)
        rng = np.random.default_rng(5)
        for g in 'fb':
            ls = np.log1p(np.exp(pn[g + '.lengthscales_unc'])) * 4.0
            pn[g + '.lengthscales_unc'] = np.log(np.expm1(ls))
            pn[g + '.zeta_mean'] = 0.5 * rng.standard_normal(pn[g + '.zeta_mean'].shape)
    p = {k: torch.tensor(v) for k, v in pn.items()}
    u, y = (torch.tensor(a) for a in syn.make_inputs(w, seed=0))
    noise = {k: torch.tensor(v) for k, v in syn.make_noise(w, seed=2).items()}
    ref = None
    for kind in ('f64', 'trsm-fp32', 'contraction-fp32', 'contraction-bf16'):
        tref.GPModel.predict = make_predict(kind)
        try:
            with torch.no_grad():
                out = tref.elbo_step(cfg, p, u, y, noise, True, want_pred=True)
        finally:
            tref.GPModel.predict = F64
        loss = float(out['loss'])
        pm, pv = out['pred_mean'].numpy(), out['pred_var'].numpy()
        if ref is None:
            ref = (loss, pm, pv)
            print('%-22s %-18s loss %.10e' % (tag, kind, loss))
            continue
        el = abs(loss - ref[0]) / abs(ref[0])
        em = np.nanmax(np.abs(pm - ref[1])) / np.max(np.abs(ref[1]))
        ev = np.nanmax(np.abs(pv - ref[2])) / np.max(np.abs(ref[2]))
        ok = np.isfinite(loss)
        print('%-22s %-18s loss %.10e  rel.err ELBO %.2e  pred_mean %.2e  pred_var %.2e%s'
              % (tag, kind, loss, el, em, ev, '' if ok else '  (not finite)'))


if __name__ == '__main__':
    torch.set_num_threads(8)
    c5, c3 = syn.WORKLOADS['C5'], syn.WORKLOADS['C3']
    run(syn.tiny(M=c5.M, dim_x=c5.dim_x, dim_u=c5.dim_u, dim_y=c5.dim_y, T=40, B=2, S=10, recog_len=8,
                 k_factor=c5.k_factor, var_x=c5.var_x, var_y=c5.var_y, gp_var=c5.gp_var, gp_len=c5.gp_len,
                 zeta_pos=c5.zeta_pos, zeta_mean=c5.zeta_mean, zeta_var=c5.zeta_var, loss_factors=c5.loss_factors),
        'C5-shaped M=300 T=40')
    run(syn.tiny(M=c5.M, dim_x=c5.dim_x, dim_u=c5.dim_u, dim_y=c5.dim_y, T=40, B=2, S=10, recog_len=8,
                 k_factor=c5.k_factor, var_x=c5.var_x, var_y=c5.var_y, gp_var=c5.gp_var, gp_len=c5.gp_len,
                 zeta_pos=c5.zeta_pos, zeta_mean=c5.zeta_mean, zeta_var=c5.zeta_var, loss_factors=c5.loss_factors),
        'C5 trained-like', trained_like=True)
    run(syn.tiny(M=c3.M, dim_x=c3.dim_x, dim_u=c3.dim_u, dim_y=c3.dim_y, T=40, B=2, S=10, recog_len=8,
                 k_factor=c3.k_factor, var_x=c3.var_x, var_y=c3.var_y, gp_var=c3.gp_var, gp_len=c3.gp_len,
                 zeta_pos=c3.zeta_pos, zeta_mean=c3.zeta_mean, zeta_var=c3.zeta_var, loss_factors=c3.loss_factors),
        'C3-shaped M=100 T=40')
    run(syn.tiny(M=c3.M, dim_x=c3.dim_x, dim_u=c3.dim_u, dim_y=c3.dim_y, T=40, B=2, S=10, recog_len=8,
                 k_factor=c3.k_factor, var_x=c3.var_x, var_y=c3.var_y, gp_var=c3.gp_var, gp_len=c3.gp_len,
                 zeta_pos=c3.zeta_pos, zeta_mean=c3.zeta_mean, zeta_var=c3.zeta_var, loss_factors=c3.loss_factors),
        'C3 trained-like', trained_like=True)
