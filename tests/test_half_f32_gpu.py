"""The forward-only variants in float32 arithmetic (reference: the `dtype` argument of cbfssmhalf.py:17 / prssm.py:17):
cbfssm_half_forward_pass_f32 / _bwd_f32 against the float64 HIP engine of the same variant -- which tests/test_half_gpu.py
and tests/test_prssm_gpu.py hold against the CPU oracle -- at every tile height.  PARITY UNPINNED like every float32
comparison here: the reference holds no float32 fixtures, the yardstick is its float64 formulation."""
import numpy as np
import pytest
import torch

from cbfssm.hip import ops
from cbfssm.hip.train_half import HipHalfGrad, HipHalfTrainStep, half_param_names

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'

CASES = [
    dict(T=11, B=3, S=4, M=12, recog_len=3),                                                  # one row block, ragged chains
    dict(T=30, B=3, S=6, M=24, dim_x=14, dim_u=7, dim_y=7, recog_len=16, k_factor=50.),      # Sarcos dims, two row blocks
    dict(T=25, B=3, S=11, M=50, dim_x=4, dim_u=1, dim_y=1, recog_len=4, k_factor=100.),      # four row blocks (C2 tile)
    dict(T=12, B=1, S=20, M=100, dim_x=14, dim_u=7, dim_y=7, recog_len=2, k_factor=50.),     # Sarcos tile
    dict(T=7, B=2, S=6, M=12, dim_x=3, dim_u=2, dim_y=3, recog_len=2),                       # no hidden dims
    dict(T=9, B=2, S=9, M=130, dim_x=6, dim_u=2, dim_y=2, recog_len=2),                      # 10 row blocks (f64: stash mode)
    dict(T=9, B=1, S=20, M=200, dim_x=14, dim_u=7, dim_y=7, recog_len=2, k_factor=50.),      # 13 row blocks: symmetric G
    dict(T=8, B=2, S=9, M=250, dim_x=4, dim_u=2, dim_y=2, recog_len=3),                      # 16 row blocks
    dict(T=10, B=2, S=9, M=300, dim_x=4, dim_u=2, dim_y=2, recog_len=5),                     # 20 row blocks
    dict(T=1, B=2, S=5, M=12, recog_len=3),                                                  # no transition at all
]


def _setup(variant, kw):
    from test_oracle import _half_setup, _prssm_setup
    return (_prssm_setup if variant == 'prssm' else _half_setup)('rnn', **kw)


@pytest.mark.parametrize('variant', ['half', 'prssm'])
@pytest.mark.parametrize('kw', CASES)
def test_half_f32_tracks_f64(variant, kw):
    """loss terms, trajectories, predictive moments within 2e-3 (float32 rounding through a T-step recurrence); every gradient
    within 2e-3 of its tensor's largest entry in the two-triangular form the automatic rule of a float32 engine runs, 1e-2 in
    the dense form (tolerances of tests/test_f32_gpu.py); a second evaluation bit-identical."""
    w, cfg, p, u, y, noise = _setup(variant, kw)
    params = {k: torch.tensor(v, device=DEV) for k, v in p.items()}
    for form in ('tri', 'dense'):
        cfg = dict(cfg)
        cfg['gp_form'] = form
        e64, e32 = HipHalfGrad(cfg, DEV, variant=variant), HipHalfGrad(cfg, DEV, variant=variant, dtype='float32')
        l64, t64, ws64 = e64.forward(params, u, y, noise, True)
        l32, t32, ws32 = e32.forward(params, u, y, noise, True)
        assert float(t32['info']) == 0.0 and e32.pack_f.gp_form() == form
        assert float(l32) == pytest.approx(float(l64), rel=5e-4, abs=1e-3)
        assert float(t32['kl_z_f']) == float(t64['kl_z_f'])             # the prior KL comes from the float64 prepare
        x64, x32 = ws64.x.cpu().numpy(), ws32.x.cpu().numpy()
        assert np.abs(x32 - x64).max() <= 2e-3 * np.abs(x64).max()
        np.testing.assert_allclose(ws32.pred_var.cpu().numpy(), ws64.pred_var.cpu().numpy(), rtol=5e-3,
                                   atol=2e-3 * float(ws64.pred_var.abs().max()))
        _, g64, _ = e64.loss_and_grads(params, u, y, noise, True)
        g64 = {k: v.cpu().numpy().copy() for k, v in g64.items()}
        l32b, g32, _ = e32.loss_and_grads(params, u, y, noise, True)
        g32 = {k: v.cpu().numpy().copy() for k, v in g32.items()}
        _, g32b, _ = e32.loss_and_grads(params, u, y, noise, True)
        assert set(g32) == set(half_param_names(cfg, variant))
        worst = 0.0
        for k in g32:
            assert np.array_equal(g32[k], g32b[k].cpu().numpy()), k
            err = np.abs(g32[k] - g64[k]).max() / (np.abs(g64[k]).max() + 1e-300)
            worst = max(worst, err)
            assert err <= (2e-3 if form == 'tri' else 1e-2), (form, k, err)
        print('\n%s float32 (%s) M=%d T=%d: worst |g32 - g64| / max|g64| %.1e' % (variant, form, w.M, w.T, worst))


@pytest.mark.parametrize('variant', ['half', 'prssm'])
def test_half_f32_tail_forms_agree(variant, monkeypatch):
    """the fused train tail (cbfssm_train_tail_half_f64 with g_mode 1 / 2) against the tensor-library restatement of the same
    tail fed with K G K: same gradients to float64 rounding."""
    for M in (50, 130, 200):
        w, cfg, p, u, y, noise = _setup(variant, dict(T=8, B=2, S=9, M=M, dim_x=6, dim_u=2, dim_y=2, recog_len=2))
        params = {k: torch.tensor(v, device=DEV) for k, v in p.items()}
        monkeypatch.delenv('CBFSSM_TORCH_TAIL', raising=False)
        _, ga, _ = HipHalfGrad(cfg, DEV, variant=variant, dtype='float32').loss_and_grads(params, u, y, noise, True)
        monkeypatch.setenv('CBFSSM_TORCH_TAIL', '1')
        eb = HipHalfGrad(cfg, DEV, variant=variant, dtype='float32')
        assert not eb.fused_tail
        _, gb, _ = eb.loss_and_grads(params, u, y, noise, True)
        monkeypatch.delenv('CBFSSM_TORCH_TAIL')
        for k in ga:
            a, b = ga[k].cpu().numpy(), gb[k].cpu().numpy()
            assert np.abs(a - b).max() <= 1e-9 * (np.abs(b).max() + 1e-300), (M, k)


def test_half_f32_train_steps_and_model_surface(tmp_path):
    """CBFSSMHALF(config, dtype='float32') / PRSSM(config, dtype='float32') through the reference's Trainer flow (HIP-graph
    train step of a float32 engine), and three Adam steps next to a float64 engine from the same start."""
    from cbfssm.datasets import make_synthetic_ds
    from cbfssm.training import Trainer
    from cbfssm.model import CBFSSMHALF, PRSSM
    from cbfssm.hip.train import TFAdam
    w, cfg, p, u, y, noise = _setup('half', dict(T=20, B=3, S=8, M=20, recog_len=4))
    losses = {}
    for dt in ('float64', 'float32'):
        eng = HipHalfGrad(cfg, DEV, dtype=dt)
        opt = TFAdam({k: torch.tensor(p[k], device=DEV) for k in half_param_names(cfg)}, 0.01)
        step = HipHalfTrainStep(eng, opt)
        losses[dt] = [float(step.step(u, y, noise, True)) for _ in range(3)]
    for a, b in zip(losses['float64'], losses['float32']):
        assert b == pytest.approx(a, rel=2e-3)
    ds_sel = make_synthetic_ds(dim_u=1, dim_y=1, n_train=400, n_test=160, seed=2)
    dim_x = 3
    base = {'ds': ds_sel, 'batch_size': 8, 'shuffle': 10000, 'seed': 7, 'dim_x': dim_x, 'ind_pnt_num': 20,
            'samples': 10, 'learning_rate': 0.05, 'loss_factors': np.asarray([1., 0.]), 'k_factor': 5.,
            'recog_len': 8, 'zeta_pos': 2., 'zeta_mean': 0.05 ** 2, 'zeta_var': 0.01 ** 2,
            'var_x': np.asarray([0.002 ** 2] * dim_x), 'var_y': np.asarray([1. ** 2] * ds_sel.dim_y),
            'gp_var': 0.5 ** 2, 'gp_len': 2.}
    for cls, extra in ((CBFSSMHALF, {}), (PRSSM, {'recog_model': 'rnn', 'loss_factors': np.asarray([0.1, 0.])})):
        c = dict(base)
        c.update(extra)
        model = cls(c, dtype='float32')
        trainer = Trainer(model, str(tmp_path / cls.__name__))
        trainer.train(ds_sel(40, 20), 3)
        assert trainer.train_all[-1] < trainer.train_all[0] and all(np.isfinite(trainer.test_all))
        assert model._engine.f32
