"""float32-arithmetic forward evaluation (cbfssm_*_f32: v_mfma_f32_16x16x4_f32, Cholesky kept in float64 as the
reference's float32 models do, gp_tf.py:57-65) against the float64 kernels and the oracle: it is a reduced-precision
path, the tolerances say how much it loses -- BASELINE.json configs[4] asks for exactly that sweep."""
import dataclasses
import json
import os
import numpy as np
import pytest
import torch

from cbfssm import synthetic as syn
from cbfssm.hip import ops

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.mark.parametrize('M,dim_x,dim_u,dim_y', [(12, 5, 2, 2), (20, 4, 1, 1), (50, 4, 1, 1), (100, 14, 7, 7),
                                                (130, 9, 3, 2), (200, 14, 7, 7), (250, 4, 2, 2), (300, 4, 2, 2)])
def test_gp_predict_f32_every_tile_height(M, dim_x, dim_u, dim_y):
    from test_hip_parity import _gp_args
    w = syn.tiny(M=M, dim_x=dim_x, dim_u=dim_u, dim_y=dim_y)
    p = syn.perturb_params(syn.make_params(w, seed=M))
    rng = np.random.default_rng(7)
    for g, Do, form in (('f', dim_x, 'dense'), ('b', dim_x - dim_y, 'dense'), ('f', dim_x, 'tri'), ('b', dim_x - dim_y, 'tri')):
        pack = ops.GPPack(M, w.D, Do, DEV, form).prepare(*[torch.tensor(a, device=DEV) for a in _gp_args(p, g)])
        for npts in (1, 16, 37):
            X = torch.tensor(rng.standard_normal((npts, w.D)) * 1.5, device=DEV)
            fm, fv = pack.predict(X)
            fm32, fv32 = pack.predict_f32(X)
            # one GP conditional in float32: a few 1e-5 of the scale of the outputs (sigma^2 for the variance) at M = 250
            np.testing.assert_allclose(fm32.cpu().numpy(), fm.cpu().numpy(), rtol=0, atol=1e-4 * float(fm.abs().max()) + 1e-7)
            np.testing.assert_allclose(fv32.cpu().numpy(), fv.cpu().numpy(), rtol=0, atol=1e-4 * float(fv.abs().max()))


@pytest.mark.parametrize('kw', [
    dict(M=100, dim_x=14, dim_u=7, dim_y=7, T=20, B=2, S=20, recog_len=4, k_factor=50., var_y=0.05 ** 2),   # Sarcos tile
    dict(M=200, dim_x=14, dim_u=7, dim_y=7, T=9, B=1, S=20, recog_len=2, k_factor=50.),                     # C4 tile
    dict(M=300, dim_x=4, dim_u=2, dim_y=2, T=60, B=2, S=9, recog_len=10, k_factor=1.),                      # C5 tile
    dict(M=20, dim_x=4, dim_u=1, dim_y=1, T=50, B=3, S=50, recog_len=16, k_factor=100., gp_len=2.),         # C1 tile
    dict(M=12, dim_x=5, dim_u=2, dim_y=2, T=11, B=3, S=4, recog_len=3, k_factor=3.),                        # ragged chains
])
@pytest.mark.parametrize('cond', [True, False])
def test_elbo_f32_tracks_f64(kw, cond):
    w = syn.tiny(loss_factors=(2., 0.7), **kw)
    cfg = w.model_config()
    p = syn.perturb_params(syn.make_params(w, seed=1), scale=0.1)
    u, y = syn.make_inputs(w, seed=0)
    noise = syn.make_noise(w, seed=2)
    e64, e32 = ops.HipElbo(cfg, DEV), ops.HipElbo(cfg, DEV, dtype='float32')
    e64.prepare(p)
    e32.prepare(p)
    o64 = e64.run(u, y, noise, condition=cond)
    o32 = e32.run(u, y, noise, condition=cond)
    a, b = o64.out.cpu().numpy(), o32.out.cpu().numpy()
    assert b[7] == 0.0
    for i, name in enumerate(('loglik', 'kl_x', 'entropy')):
        assert b[i] == pytest.approx(a[i], rel=2e-4, abs=1e-3), name
    assert b[3] == a[3] and b[4] == a[4]                    # the prior KL comes from the float64 prepare
    assert b[6] == pytest.approx(a[6], rel=2e-4)
    x64, x32 = o64.x.cpu().numpy(), o32.x.cpu().numpy()
    assert np.abs(x32 - x64).max() <= 2e-3 * np.abs(x64).max()
    np.testing.assert_allclose(o32.pred_mean.cpu().numpy(), o64.pred_mean.cpu().numpy(), rtol=0,
                               atol=2e-3 * float(o64.pred_mean.abs().max()))
    # a second evaluation is bit-identical
    out1 = o32.out.clone()
    o32 = e32.run(u, y, noise, condition=cond)
    assert torch.equal(out1, o32.out)


GRAD_CASES = [
    dict(M=12, dim_x=5, dim_u=2, dim_y=2, T=11, B=3, S=4, recog_len=3, k_factor=3.),                        # ragged chains, one row block
    dict(M=24, dim_x=14, dim_u=7, dim_y=7, T=40, B=3, S=6, recog_len=16, k_factor=50., var_y=0.05 ** 2),    # Sarcos dims, two row blocks
    dict(M=50, dim_x=4, dim_u=1, dim_y=1, T=25, B=3, S=11, recog_len=4, k_factor=100., gp_len=2.),          # four row blocks (C2 tile)
    dict(M=100, dim_x=14, dim_u=7, dim_y=7, T=20, B=2, S=20, recog_len=4, k_factor=50., var_y=0.05 ** 2),   # C3 tile (7 row blocks)
    dict(M=130, dim_x=9, dim_u=3, dim_y=2, T=14, B=2, S=9, recog_len=3, k_factor=5.),                       # 10 row blocks (f64: stash mode)
    dict(M=200, dim_x=14, dim_u=7, dim_y=7, T=9, B=1, S=20, recog_len=2, k_factor=50.),                     # C4 tile, two row blocks per wave
    dict(M=250, dim_x=4, dim_u=2, dim_y=2, T=12, B=2, S=9, recog_len=50, k_factor=1.),                      # 16 row blocks: two passes, T < R
    dict(M=300, dim_x=4, dim_u=2, dim_y=2, T=24, B=2, S=9, recog_len=5, k_factor=1.),                       # C5 tile: two passes over the time loop
]


@pytest.mark.parametrize('form', ['tri', 'dense'])
@pytest.mark.parametrize('kw', GRAD_CASES)
def test_f32_adjoint_matches_f64_adjoint_and_f32_restatement(kw, form):
    """The float32 adjoint (cbfssm_*_pass_bwd_f32; what `minimize` differentiates for CBFSSM(config, dtype=float32),
    cbfssm.py:12,273-275) at every tile height: all twelve gradients against (a) the float64 HIP adjoint on the same
    inputs and (b) the PyTorch-CPU restatement run in float32 (every tensor float32, Cholesky through float64 as
    gp_tf.py:57-65) -- PARITY UNPINNED like every oracle comparison here (no TensorFlow, no reference fixtures).
    Both GP forms: 'tri' (what the automatic rule of a float32 engine runs above cond 1e3: every product with K^-1 as two
    triangular products, the reference's order) and 'dense' (explicit K^-1).  Tolerance against float64: 2e-3 of each tensor's
    largest entry (float32 rounding through a T-step recurrence and its reverse sweep; achieved values are printed), 1e-2 for
    the dense form (it loses cond(K_mm) eps_32 per product: 2.1e-3 at the M = 50 case, cond 1e5 -- which the automatic rule
    of a float32 engine never runs dense), and the two-triangular kernel must be as
    close to float64 as the float32 CPU restatement is, within a factor 20, on every case -- ill-conditioned ones included: the
    kernel accumulates (K^-1 A2bar) A2^T, so nothing multiplies a float32 accumulator by K^-1 from both sides (with the
    d loss / d K^-1 accumulator of the float64 kernels it was 1.6e-3 at cond 4e4 against 2.5e-5 for the restatement)."""
    from cbfssm.hip import train
    from oracle import cbfssm_torch_ref as tref
    w = syn.tiny(loss_factors=(2., 0.7), **kw)
    cfg = w.model_config()
    cfg['gp_form'] = form
    p = syn.perturb_params(syn.make_params(w, seed=1), scale=0.1)
    u, y = syn.make_inputs(w, seed=0)
    noise = syn.make_noise(w, seed=2)
    params = {k: torch.tensor(v, device=DEV) for k, v in p.items()}
    l64, g64, _ = train.HipElboGrad(cfg, DEV).loss_and_grads(params, u, y, noise)
    g64 = {k: v.cpu().numpy().copy() for k, v in g64.items()}
    e32 = train.HipElboGrad(cfg, DEV, dtype='float32')
    l32, g32, t32 = e32.loss_and_grads(params, u, y, noise)
    assert float(t32['info']) == 0.0 and e32.pack_f.gp_form() == form
    g32 = {k: v.cpu().numpy().copy() for k, v in g32.items()}
    # a second evaluation is bit-identical (fixed-order reductions, no atomics)
    _, g32b, _ = e32.loss_and_grads(params, u, y, noise)
    for k in train.PARAM_NAMES:
        assert np.array_equal(g32[k], g32b[k].cpu().numpy()), k
    _, gt32 = tref.loss_and_grads(cfg, p, u, y, noise, True, dtype=torch.float32)
    assert float(l32) == pytest.approx(float(l64), rel=2e-4)
    from cbfssm.hip import lib as _lib
    cond = max(float(e32.pack_f.scal[_lib.SCAL_COND]), float(e32.pack_b.scal[_lib.SCAL_COND]))
    worst = {}
    for k in train.PARAM_NAMES:
        scale = np.abs(g64[k]).max()
        e_hip = np.abs(g32[k] - g64[k]).max() / scale
        e_cpu = np.abs(gt32[k].astype(np.float64) - g64[k]).max() / scale
        worst[k] = (e_hip, e_cpu)
        assert e_hip <= (2e-3 if form == 'tri' else 1e-2), (k, e_hip, e_cpu)
        if form == 'tri':
            assert e_hip <= 20.0 * max(e_cpu, 1e-6), (k, e_hip, e_cpu)
    print('\nfloat32 adjoint (%s form) M=%d T=%d cond %.1e: worst |g32 - g64| / max|g64| HIP %.1e, CPU float32 restatement %.1e'
          % (form, w.M, w.T, cond, max(v[0] for v in worst.values()), max(v[1] for v in worst.values())))


def test_f32_train_steps_track_f64():
    """Three Adam steps (HipTrainStep: HIP graph, TF-1.8 update rule) of a float32 engine next to a float64 one from the same
    initial values and noise: the parameters stay together to float32 accuracy."""
    from cbfssm.hip.train import HipTrainStep, PARAM_NAMES
    w = syn.tiny(M=20, T=13, B=2, S=8)
    cfg = w.model_config()
    p = syn.perturb_params(syn.make_params(w))
    u, y = syn.make_inputs(w)
    params = {k: torch.tensor(v, device=DEV) for k, v in p.items()}
    s64 = HipTrainStep(cfg, params, DEV)
    s32 = HipTrainStep(cfg, params, DEV, dtype='float32')
    for k in range(3):
        noise = syn.make_noise(w, seed=10 + k)
        a = float(s64.step(u, y, noise))
        b = float(s32.step(u, y, noise))
        assert b == pytest.approx(a, rel=1e-3)
    for k in PARAM_NAMES:
        x64, x32 = s64.params[k].cpu().numpy(), s32.params[k].cpu().numpy()
        assert np.abs(x32 - x64).max() <= 2e-3 * max(np.abs(x64).max(), 1e-3), k


def test_f32_model_surface(tmp_path):
    """CBFSSM(config, dtype='float32') (cbfssm.py:12): loss, predictions and `sess.run(model.train)` through the float32
    passes and the float32 adjoint"""
    from cbfssm.datasets import make_synthetic_ds
    from cbfssm.model import CBFSSM
    from cbfssm.model.session import Session
    ds_sel = make_synthetic_ds(dim_u=1, dim_y=1, n_train=200, n_test=80, seed=1)
    dim_x = 3
    cfg = {'ds': ds_sel, 'batch_size': 4, 'shuffle': 100, 'seed': 3, 'dim_x': dim_x, 'ind_pnt_num': 20, 'samples': 10,
           'learning_rate': 0.05, 'loss_factors': np.asarray([1., 0.]), 'k_factor': 5., 'recog_len': 8, 'zeta_pos': 2.,
           'zeta_mean': 0.05 ** 2, 'zeta_var': 0.01 ** 2, 'var_x': np.asarray([0.002 ** 2] * dim_x),
           'var_y': np.asarray([1. ** 2] * dim_x), 'gp_var': 0.5 ** 2, 'gp_len': 2.}
    ds = ds_sel(40, 20)
    res = {}
    for dt in ('float64', 'float32'):
        m = CBFSSM(dict(cfg), dtype=dt)
        with m.graph.as_default(), Session() as sess:
            sess.run(m.init)
            m.load_ds(sess, ds.train_in_batch, ds.train_out_batch)
            res[dt] = m.run(sess, (m.loss, m.pred_mean), {m.condition: True})
            # one epoch of training (training/trainer.py:39-41): the losses of the two dtypes stay together
            m.load_ds(sess, ds.train_in_batch, ds.train_out_batch)
            res[dt + '_train'] = m.run(sess, (m.train, m.loss), {m.condition: True})[1]
    assert np.all(np.isfinite(res['float32_train']))
    np.testing.assert_allclose(res['float32_train'], res['float64_train'], rtol=5e-3)
    np.testing.assert_allclose(res['float32'][0], res['float64'][0], rtol=1e-3)
    np.testing.assert_allclose(res['float32'][1], res['float64'][1], rtol=0, atol=5e-3 * np.abs(res['float64'][1]).max())


def test_precision_sweep_c5_shape_on_the_gpu():
    """BASELINE.json configs[4]: the ELBO / predictive-moment error of float32 arithmetic against float64 on the
    RoboMove-shaped problem (M = 300, T = 1000, S = 50, recog_len = 50; B = 8 here) -- and of bf16 operands in the K^-1 K
    contraction with float32 accumulation -- at the run-script initial values and at trained-like parameters.  The numbers go to gpurun_out/precision_sweep_gpu.json (quoted in DESIGN.md)."""
    w = dataclasses.replace(syn.WORKLOADS['C5'], B=8)
    cfg = w.model_config()
    u, y = syn.make_inputs(w, seed=0)
    noise = syn.make_noise(w, seed=2)
    report = {}
    for tag, p in (('initial', syn.perturb_params(syn.make_params(w, seed=1), scale=0.1)),
                   ('trained_like_ls_x2', syn.trained_like_params(w, ls_mult=2.0, zeta_mean=0.05)),
                   ('trained_like_ls_x3', syn.trained_like_params(w, ls_mult=3.0, zeta_mean=0.05))):
        e64 = ops.HipElbo(cfg, DEV)
        e64.prepare(p)
        o64 = e64.run(u, y, noise, condition=True)
        a = o64.out.cpu().numpy()
        pm64, pv64 = o64.pred_mean.cpu().numpy(), o64.pred_var.cpu().numpy()
        rep = {'cond_f': float(e64.pack_f.scal[4]), 'cond_b': float(e64.pack_b.scal[4])}
        for dt in ('float32', 'bfloat16'):      # bfloat16: bf16 operands of the K^-1 K contraction, float32 accumulation
            e32 = ops.HipElbo(cfg, DEV, dtype=dt)
            e32.prepare(p)
            o32 = e32.run(u, y, noise, condition=True)
            b = o32.out.cpu().numpy()
            rep[dt] = {'loss_rel': float(abs(b[6] - a[6]) / abs(a[6])),
                       'pred_mean_relmax': float(np.abs(o32.pred_mean.cpu().numpy() - pm64).max() / np.abs(pm64).max()),
                       'pred_var_rel': float((np.abs(o32.pred_var.cpu().numpy() - pv64) / pv64).max())}
            print('\nC5-shape (B=8) %s vs float64, %s: cond f %.1e b %.1e | loss %.1e pred_mean %.1e pred_var %.1e'
                  % (dt, tag, rep['cond_f'], rep['cond_b'], rep[dt]['loss_rel'], rep[dt]['pred_mean_relmax'],
                     rep[dt]['pred_var_rel']))
            assert np.isfinite(b[6])
        report[tag] = rep
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
    os.makedirs(out, exist_ok=True)
    json.dump(report, open(os.path.join(out, 'precision_sweep_gpu.json'), 'w'), indent=1)
    assert report['initial']['float32']['loss_rel'] <= 1e-4
    assert report['initial']['bfloat16']['loss_rel'] <= 1e-2


def test_f32_kept_records_above_2_31_floats():
    """The float32 passes keep every step's [A2 | kernel tile] registers for the adjoint (cbfssm_saved_a2_f32_elems floats).
    At M = 300, T = 200, B = 192, S = 50 the backward-run buffer has 2.46e9 floats (9.8 GB): above 2^31 elements, where a
    32-bit count or offset anywhere between the size query and the kernels is a GPU memory fault (it was one, round 4: the
    ctypes binding had declared the size query `int`).  Same loss, gradients to float32 rounding, as the recomputing path."""
    import dataclasses
    import os
    from cbfssm.hip.train import HipElboGrad, PARAM_NAMES
    w = dataclasses.replace(syn.WORKLOADS['C5'], T=200, B=192)
    cfg = w.model_config()
    p = {k: torch.tensor(v, device=DEV) for k, v in syn.make_params(w, seed=1).items()}
    g = torch.Generator(device=DEV)
    g.manual_seed(3)
    u = torch.randn(w.B, w.T, w.dim_u, dtype=torch.float64, device=DEV, generator=g)
    y = torch.randn(w.B, w.T, w.dim_y, dtype=torch.float64, device=DEV, generator=g)
    N, T = w.N, w.T
    noise = {k: torch.randn(n, dtype=torch.float64, device=DEV, generator=g)
             for k, n in (('hid_b', 2 * T * N), ('eps_b', 2 * T * N), ('eps_f', (T - 1) * N))}
    res = {}
    for mode in ('recompute', 'kept'):
        if mode == 'recompute':
            os.environ['CBFSSM_F32_NO_TILES'] = '1'
        else:
            os.environ.pop('CBFSSM_F32_NO_TILES', None)
        try:
            eng = HipElboGrad(cfg, DEV, dtype='float32')
            loss, grads, _ = eng.loss_and_grads(p, u, y, noise)
            torch.cuda.synchronize()
            assert (eng.last_ws.a2s_b is not None) == (mode == 'kept')
            if mode == 'kept':
                assert eng.last_ws.a2s_b.numel() * 2 > 2 ** 31
            res[mode] = (float(loss), {k: grads[k].clone() for k in PARAM_NAMES})
        finally:
            os.environ.pop('CBFSSM_F32_NO_TILES', None)
        del eng
        torch.cuda.empty_cache()
    assert res['kept'][0] == pytest.approx(res['recompute'][0], rel=1e-12)
    for k in PARAM_NAMES:
        a, b = res['kept'][1][k], res['recompute'][1][k]
        assert float((a - b).abs().max()) <= 5e-6 * float(b.abs().max()) + 1e-30, k
