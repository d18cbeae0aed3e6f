"""Generate tests/golden/*.npz from the CPU oracle (TEST INFRASTRUCTURE; see oracle/cbfssm_oracle.py header).

PARITY UNPINNED: the vectors are outputs of this repo's own float64 restatement of the reference, not of the
reference itself (TensorFlow 1.8 is not installable here; the reference holds no fixtures).  Run from the repo root:

    python oracle/make_golden.py
"""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'cbf-ssm_amd')]

from cbfssm import synthetic as syn          # noqa: E402
from oracle import cbfssm_oracle as orc      # noqa: E402
from oracle import cbfssm_torch_ref as tref  # noqa: E402

CASES = {
    # generic small case, every dimension distinct, both loss factors active, windows shorter than T
    'tiny': syn.tiny(),
    # shrunken Sarcos: real dims (7/7/14, D=21), k_factor 50, small M/T/B/S
    'mini_sarcos': syn.Workload('mini_sarcos', dim_u=7, dim_y=7, dim_x=14, M=24, T=40, B=3, S=6, recog_len=16,
                                k_factor=50., loss_factors=(6., 0.3), var_y=0.05 ** 2),
    # shrunken small-scale (Actuator-like 1/1/4, D=5), M not a multiple of 16, ragged T vs 2R
    'mini_smallscale': syn.Workload('mini_smallscale', dim_u=1, dim_y=1, dim_x=4, M=20, T=37, B=5, S=7,
                                    recog_len=16, k_factor=100., loss_factors=(0.5, 0.1), gp_len=2.),
}


def main():
    out_dir = os.path.join(ROOT, 'tests', 'golden')
    os.makedirs(out_dir, exist_ok=True)
    for name, w in CASES.items():
        cfg = w.model_config()
        p = syn.perturb_params(syn.make_params(w, seed=1))
        u, y = syn.make_inputs(w, seed=0)
        noise = syn.make_noise(w, seed=2)
        blob = {'workload_' + k: np.asarray(v) for k, v in syn.workload_dict(w).items() if k != 'name'}
        blob.update({'param_' + k: v for k, v in p.items()})
        blob.update({'u': u, 'y': y})
        blob.update({'noise_' + k: v for k, v in noise.items()})
        for cond in (True, False):
            tag = 'c1_' if cond else 'c0_'
            trace = {}
            res = orc.elbo_step(cfg, p, u, y, noise, cond, trace)
            for k in ('loss', 'loglik', 'kl_x', 'entropy', 'kl_z_f', 'kl_z_b', 'pred_mean', 'pred_var',
                      'x_final', 'y_tilde'):
                blob[tag + k] = np.asarray(res[k])
            if cond:
                # one GP call of each pass, for the stand-alone gp_predict kernel
                blob['f_fmean_t0'] = trace['f_fmean'][0]
                blob['f_fvar_t0'] = trace['f_fvar'][0]
            scal, grads = tref.loss_and_grads(cfg, p, u, y, noise, cond)
            assert abs(scal['loss'] - res['loss']) <= 1e-9 * abs(res['loss'])
            blob.update({tag + 'grad_' + k: g for k, g in grads.items()})
        path = os.path.join(out_dir, name + '.npz')
        np.savez_compressed(path, **blob)
        print(name, os.path.getsize(path) // 1024, 'KiB', 'loss', blob['c1_loss'])


# Full-length recurrences at reduced batch: the shapes BASELINE.json quotes its tolerance on (T = 250 / 1000 / 100 with
# the configured S, M, recog_len), B cut to what the oracle and the autograd restatement finish in seconds.
FULL_CASES = {
    'full_C2': dict(base='C2', B=4),      # Actuator: M=50  T=100  S=50 R=16
    'full_C3': dict(base='C3', B=2),      # Sarcos:   M=100 T=250  S=20 R=16
    'full_C4': dict(base='C4', B=2),      # Sarcos:   M=200 T=250  S=20 R=16
    'full_C5': dict(base='C5', B=1),      # RoboMove: M=300 T=1000 S=50 R=50
}
T_STRIDE = 8      # x_final / y_tilde are stored at every 8th step and at the last one (the recurrence carries every
                  # earlier error into those; pred_mean / pred_var are stored in full)


def main_full():
    import dataclasses
    out_dir = os.path.join(ROOT, 'tests', 'golden')
    for name, spec in FULL_CASES.items():
        w = dataclasses.replace(syn.WORKLOADS[spec['base']], B=spec['B'], name=name)
        cfg = w.model_config()
        p = syn.perturb_params(syn.make_params(w, seed=1), scale=0.1)
        u, y = syn.make_inputs(w, seed=0)
        noise = syn.make_noise(w, seed=2)
        blob = {'workload_' + k: np.asarray(v) for k, v in syn.workload_dict(w).items() if k != 'name'}
        blob.update({'param_' + k: v for k, v in p.items()})
        blob.update({'u': u, 'y': y})
        blob.update({'noise_' + k: v for k, v in noise.items()})
        tsel = np.unique(np.concatenate((np.arange(0, w.T, T_STRIDE), [w.T - 1])))
        blob['t_sel'] = tsel
        for cond in (True, False):
            tag = 'c1_' if cond else 'c0_'
            res = orc.elbo_step(cfg, p, u, y, noise, cond)
            for k in ('loss', 'loglik', 'kl_x', 'entropy', 'kl_z_f', 'kl_z_b', 'pred_mean', 'pred_var'):
                blob[tag + k] = np.asarray(res[k])
            blob[tag + 'x_final_sel'] = np.asarray(res['x_final'][:, tsel])
            blob[tag + 'y2_sel'] = np.asarray(res['y_tilde'][:, tsel][..., w.dim_y:])
            if cond:
                scal, grads = tref.loss_and_grads(cfg, p, u, y, noise, cond)
                assert abs(scal['loss'] - res['loss']) <= 1e-9 * abs(res['loss'])
                blob.update({tag + 'grad_' + k: g for k, g in grads.items()})
        path = os.path.join(out_dir, name + '.npz')
        np.savez_compressed(path, **blob)
        print(name, os.path.getsize(path) // 1024, 'KiB', 'loss', blob['c1_loss'], flush=True)


def main_half():
    """CBFSSMHALF fixture (cbfssm/model/cbfssmhalf.py): GRU recognition model, both `condition` settings."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from test_oracle import _half_setup
    w, cfg, p, u, y, noise = _half_setup('rnn', T=13, B=3, S=5, M=14, recog_len=4)
    blob = {'workload_' + k: np.asarray(v) for k, v in syn.workload_dict(w).items() if k != 'name'}
    blob.update({'param_' + k: v for k, v in p.items()})
    blob.update({'u': u, 'y': y, 'noise_eps_f': noise['eps_f'], 'var_y_cfg': cfg['var_y']})
    for cond in (True, False):
        tag = 'c1_' if cond else 'c0_'
        res = orc.CBFSSMHALFOracle(cfg, p).run(u, y, noise, cond)
        for k in ('loss', 'loglik', 'kl_x', 'kl_z_f', 'pred_mean', 'pred_var', 'x_final'):
            blob[tag + k] = np.asarray(res[k])
        loss, grads = tref.half_loss_and_grads(cfg, p, u, y, noise, cond)
        assert abs(loss - res['loss']) <= 1e-9 * abs(res['loss'])
        blob.update({tag + 'grad_' + k: g for k, g in grads.items()})
    path = os.path.join(ROOT, 'tests', 'golden', 'half_tiny.npz')
    np.savez_compressed(path, **blob)
    print('half_tiny', os.path.getsize(path) // 1024, 'KiB', 'loss', blob['c1_loss'])


# Trained-like, ill-conditioned parameters (cbfssm.synthetic.trained_like_params) through the full C3 recurrence: the
# oracle's outputs, the reproducibility floor of the reference formulation (numpy/LAPACK oracle vs the PyTorch-CPU
# restatement of the same two triangular solves) and reverse-mode gradients of the restatement.  These take minutes of
# CPU time; the GPU tests read them from tests/golden/trained_*.npz (inputs are regenerated from the seeds; a checksum of
# the parameters guards the generators).
TRAINED_SWEEP = [8, 16, 32, 64, 128, 256]
TRAINED_GRADS = [(64, 'C3', 250), (32, 'C4', 60)]


def trained_case(base, ls_mult, T=None):
    import dataclasses
    kw = dict(B=2)
    if T is not None:
        kw['T'] = T
    w = dataclasses.replace(syn.WORKLOADS[base], **kw)
    p = syn.trained_like_params(w, ls_mult=float(ls_mult), zeta_mean=0.1)
    u, y = syn.make_inputs(w, seed=0)
    noise = syn.make_noise(w, seed=2)
    return w, p, u, y, noise


def param_checksum(p):
    return float(sum(float(np.sum(v * np.cos(np.arange(v.size).reshape(v.shape)))) for _, v in sorted(p.items())))


def main_trained():
    import torch
    out_dir = os.path.join(ROOT, 'tests', 'golden')
    blob = {'sweep': np.asarray(TRAINED_SWEEP)}
    tsel = None
    for m in TRAINED_SWEEP:
        w, p, u, y, noise = trained_case('C3', m)
        cfg = w.model_config()
        if tsel is None:
            tsel = np.unique(np.concatenate((np.arange(0, w.T, T_STRIDE), [w.T - 1])))
            blob['t_sel'] = tsel
        ref = orc.elbo_step(cfg, p, u, y, noise, True)
        with torch.no_grad():
            o = tref.elbo_step(cfg, {k: torch.tensor(v) for k, v in p.items()}, torch.tensor(u), torch.tensor(y),
                               {k: torch.tensor(v) for k, v in noise.items()}, True, want_pred=True)
        tag = 'x%d_' % m
        for k in ('loss', 'loglik', 'kl_x', 'entropy', 'kl_z_f', 'kl_z_b', 'pred_mean', 'pred_var'):
            blob[tag + k] = np.asarray(ref[k])
        blob[tag + 'x_final_sel'] = np.asarray(ref['x_final'][:, tsel])
        blob[tag + 'y2_sel'] = np.asarray(ref['y_tilde'][:, tsel][..., w.dim_y:])
        blob[tag + 'floor_loss'] = abs(float(o['loss']) - float(ref['loss'])) / abs(float(ref['loss']))
        blob[tag + 'floor_pred_mean'] = np.abs(o['pred_mean'].numpy() - ref['pred_mean']).max() / np.abs(ref['pred_mean']).max()
        blob[tag + 'floor_pred_var'] = (np.abs(o['pred_var'].numpy() - ref['pred_var']) / np.abs(ref['pred_var'])).max()
        blob[tag + 'cond_f'] = syn.kmm_condition(p, 'f')
        blob[tag + 'cond_b'] = syn.kmm_condition(p, 'b')
        blob[tag + 'param_checksum'] = param_checksum(p)
        print('sweep x%d' % m, 'cond %.1e' % blob[tag + 'cond_f'], 'floor', blob[tag + 'floor_loss'],
              blob[tag + 'floor_pred_mean'], blob[tag + 'floor_pred_var'], flush=True)
    for m, base, T in TRAINED_GRADS:
        w, p, u, y, noise = trained_case(base, m, T)
        scal, gref = tref.loss_and_grads(w.model_config(), p, u, y, noise, True)
        tag = 'g_%s_x%d_' % (base, m)
        blob[tag + 'loss'] = scal['loss']
        blob[tag + 'param_checksum'] = param_checksum(p)
        blob.update({tag + 'grad_' + k: g for k, g in gref.items()})
        print('grad', base, m, T, scal['loss'], flush=True)
    path = os.path.join(out_dir, 'trained_C3.npz')
    np.savez_compressed(path, **blob)
    print('trained_C3', os.path.getsize(path) // 1024, 'KiB')


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'full':
        main_full()
    elif len(sys.argv) > 1 and sys.argv[1] == 'trained':
        main_trained()
    else:
        main()
        main_half()
        main_full()
