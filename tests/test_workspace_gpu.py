"""Saved-tile pool of an engine (cbfssm.hip.ops.TilePool): a ragged final mini-batch and an eval pass at another batch
size reuse the HBM of the largest shape -- the saved-A2 adjoint stays on for every shape and the footprint does not
double; a shape over the budget falls back to the recompute adjoint and says so."""
import dataclasses
import warnings
import numpy as np
import pytest
import torch

from cbfssm import synthetic as syn
from cbfssm.hip import train

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _batch(w, B, seed):
    wb = dataclasses.replace(w, B=B)
    u, y = syn.make_inputs(wb, seed=seed)
    return u, y, syn.make_noise(wb, seed=seed + 1)


def test_ragged_final_batch_shares_the_saved_tiles_at_c3_shape():
    w = syn.WORKLOADS['C3']
    cfg = w.model_config()
    p = syn.perturb_params(syn.make_params(w, seed=1), scale=0.1)
    st = train.HipTrainStep(cfg, {k: torch.tensor(v, device=DEV) for k, v in p.items()}, DEV)
    eng = st.engine
    torch.cuda.synchronize()
    l_full = float(st.step(*_batch(w, w.B, 0)))
    torch.cuda.synchronize()
    pool0, mem0 = eng.tile_pool.bytes(), torch.cuda.memory_allocated()
    assert pool0 > 2 * 2 ** 30                                     # C3 keeps 6.8 GB of [A2 | K] records (3.4 GB of A2 tiles alone)
    ws_full = eng.last_ws
    assert ws_full.a2s_f is not None and ws_full.a2s_b is not None
    with warnings.catch_warnings():
        warnings.simplefilter('error')                             # no growth, no fallback: nothing may warn
        l_rag = float(st.step(*_batch(w, 77, 2)))                  # the partial last batch of an epoch
        l_eval, _, _ = eng.forward(st.params, *_batch(w, 64, 4))   # Trainer's test pass at another batch size
        l_full2 = float(st.step(*_batch(w, w.B, 0)))
    torch.cuda.synchronize()
    ws_rag = eng._ws[(77, w.T)]
    assert ws_rag.a2s_f is not None and ws_rag.a2s_b is not None   # the saved-A2 adjoint stays on
    assert ws_rag.a2s_f.data_ptr() == ws_full.a2s_f.data_ptr()     # ... in the same HBM
    assert eng.tile_pool.bytes() == pool0 and not eng.tile_pool.log
    assert torch.cuda.memory_allocated() - mem0 < 0.5 * pool0      # the other per-shape buffers only
    assert np.isfinite([l_full, l_rag, float(l_eval), l_full2]).all()


def test_ragged_batch_gradient_is_the_same_with_shared_tiles():
    """the gradient of a small batch evaluated after a large one (tiles viewed into the large pool) equals the gradient of
    the same batch on a fresh engine"""
    w = syn.tiny(M=100, dim_x=14, dim_u=7, dim_y=7, T=24, B=6, S=20, recog_len=4, k_factor=50., var_y=0.05 ** 2)
    cfg = w.model_config()
    p = {k: torch.tensor(v, device=DEV) for k, v in syn.perturb_params(syn.make_params(w, seed=1), scale=0.1).items()}
    big, small = _batch(w, 6, 0), _batch(w, 2, 5)
    e1 = train.HipElboGrad(cfg, DEV)
    e1.loss_and_grads(p, *big)
    l1, g1, _ = e1.loss_and_grads(p, *small)
    g1 = {k: v.clone() for k, v in g1.items()}
    e2 = train.HipElboGrad(cfg, DEV)
    l2, g2, _ = e2.loss_and_grads(p, *small)
    assert float(l1) == float(l2)
    for k in train.PARAM_NAMES:
        assert torch.equal(g1[k], g2[k]), k


def test_shape_over_the_budget_falls_back_loudly(monkeypatch):
    monkeypatch.setenv('CBFSSM_A2S_MAX_GB', '0.0001')      # 107 KB: the tiles of this shape need 770 KB
    w = syn.tiny(M=20, T=16, B=8, S=8)
    cfg = w.model_config()
    p = {k: torch.tensor(v, device=DEV) for k, v in syn.make_params(w).items()}
    eng = train.HipElboGrad(cfg, DEV)
    with pytest.warns(UserWarning, match='recomputes A2'):
        loss, grads, _ = eng.loss_and_grads(p, *_batch(w, 8, 0))
    assert eng.last_ws.a2s_f is None and eng.tile_pool.log
    monkeypatch.delenv('CBFSSM_A2S_MAX_GB')
    ref = train.HipElboGrad(cfg, DEV)
    loss_r, grads_r, _ = ref.loss_and_grads(p, *_batch(w, 8, 0))
    assert ref.last_ws.a2s_f is not None
    assert float(loss) == float(loss_r)
    for k in train.PARAM_NAMES:
        np.testing.assert_allclose(grads[k].cpu().numpy(), grads_r[k].cpu().numpy(), rtol=1e-9,
                                   atol=1e-12 * float(grads_r[k].abs().max()))


@pytest.mark.parametrize('M', [20, 100])
def test_kept_kernel_tiles_give_the_gradient_of_the_recomputing_adjoint(M, monkeypatch):
    """Tile heights up to seven row blocks keep the kernel tile K = k(Z, x_t) of every step next to its A2 tile and the
    adjoint reads it (rev_kernel<..., KSV>); with CBFSSM_NO_SAVE_K=1 the records hold the A2 tile only and the adjoint
    rebuilds K from the saved trajectory.  Same inputs, same arithmetic for K: the two adjoints must agree to rounding."""
    import ctypes as C
    from cbfssm.hip import lib as _l
    w = syn.tiny(M=M, dim_x=14, dim_u=7, dim_y=7, T=40, B=3, S=20, recog_len=8, k_factor=50., var_y=0.05 ** 2)
    cfg = w.model_config()
    p = {k: torch.tensor(v, device=DEV) for k, v in syn.perturb_params(syn.make_params(w, seed=3), scale=0.1).items()}
    batch = _batch(w, w.B, 7)
    e1 = train.HipElboGrad(cfg, DEV)
    l1, g1, _ = e1.loss_and_grads(p, *batch)
    g1 = {k: v.clone() for k, v in g1.items()}
    n_kept = e1.tile_pool.bytes()
    monkeypatch.setenv('CBFSSM_NO_SAVE_K', '1')
    e2 = train.HipElboGrad(cfg, DEV)
    l2, g2, _ = e2.loss_and_grads(p, *batch)
    assert e2.tile_pool.bytes() * 2 == n_kept                      # the record is [A2 | K] resp. [A2]
    assert float(l1) == float(l2)                                  # the forward evaluation is the same code
    for k in g1:
        a, b = g1[k].cpu().numpy(), g2[k].cpu().numpy()
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-12 * max(np.abs(b).max(), 1e-300), err_msg=k)
