"""Two ranks sharing the one GPU of the test box (gloo moves the flat buffer through host memory): the sharded
train step -- per-rank kernels, ONE all-reduce, prior KL added once -- equals the single-process evaluation of the
global batch bit-for-bit up to summation order (rel 1e-10)."""
import datetime
import os
import socket
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _case():
    from cbfssm import synthetic as syn
    w = syn.tiny(M=20, T=21, B=6, S=8, recog_len=4)
    return w, syn.perturb_params(syn.make_params(w)), syn.make_inputs(w), syn.make_noise(w)


def _worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    try:
        from cbfssm.hip import train
        from cbfssm.hip.dist_utils import shard_range
        w, p, (u, y), noise = _case()
        lo, hi = shard_range(w.B, rank, world)
        nz = {'hid_b': noise['hid_b'][:, :, lo:hi], 'eps_b': noise['eps_b'][:, :, lo:hi],
              'eps_f': noise['eps_f'][:, lo:hi]}
        nz = {k: np.ascontiguousarray(v) for k, v in nz.items()}
        eng = train.HipElboGrad(w.model_config(), 'cuda:0', dist)
        params = {k: torch.tensor(v, device='cuda:0') for k, v in p.items()}
        loss, grads, _ = eng.loss_and_grads(params, u[lo:hi], y[lo:hi], nz)
        loss_e, _, _ = eng.forward(params, u[lo:hi], y[lo:hi], nz)
        out[rank] = (float(loss), float(loss_e), {k: g.cpu().numpy() for k, g in grads.items()})
    finally:
        dist.destroy_process_group()


def test_sharded_step_equals_global_batch():
    from cbfssm.hip import train
    w, p, (u, y), noise = _case()
    eng = train.HipElboGrad(w.model_config(), 'cuda:0')
    params = {k: torch.tensor(v, device='cuda:0') for k, v in p.items()}
    loss, grads, _ = eng.loss_and_grads(params, u, y, noise)
    ref = (float(loss), {k: g.cpu().numpy() for k, g in grads.items()})
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    for rank in (0, 1):
        l, le, g = out[rank]
        assert l == pytest.approx(ref[0], rel=1e-10)
        assert le == pytest.approx(ref[0], rel=1e-10)
        for k in train.PARAM_NAMES:
            np.testing.assert_allclose(g[k], ref[1][k], rtol=1e-8, atol=1e-9 * np.abs(ref[1][k]).max())


def _step_worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    try:
        from cbfssm.hip import train
        from cbfssm.hip.dist_utils import shard_range
        w, p, (u, y), noise = _case()
        lo, hi = shard_range(w.B, rank, world)
        res = {}
        for graph in (True, False):
            st = train.HipTrainStep(w.model_config(), {k: torch.tensor(v, device='cuda:0') for k, v in p.items()},
                                    'cuda:0', dist, graph=graph)
            assert st.use_graph == graph
            losses = []
            for i in range(3):                                    # fresh inputs every step: the graphs' static copies
                nz = {'hid_b': noise['hid_b'][:, :, lo:hi] * (1 + i), 'eps_b': noise['eps_b'][:, :, lo:hi],
                      'eps_f': noise['eps_f'][:, lo:hi] * (1 - 0.1 * i)}
                nz = {k: np.ascontiguousarray(v) for k, v in nz.items()}
                losses.append(float(st.step(u[lo:hi] * (1 + 0.01 * i), y[lo:hi], nz)))
            res[graph] = (losses, {k: v.detach().cpu().numpy().copy() for k, v in st.params.items()})
        out[rank] = res
    finally:
        dist.destroy_process_group()


def test_data_parallel_step_graphs_equal_eager():
    """With a process group HipTrainStep replays two HIP graphs around the eager all-reduce; losses and parameters after
    three Adam steps equal the eager data-parallel step bit for bit, and the ranks stay identical."""
    from cbfssm.hip import train
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_step_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    for rank in (0, 1):
        lg, pg = out[rank][True]
        le, pe = out[rank][False]
        assert lg == le
        for k in train.PARAM_NAMES:
            np.testing.assert_array_equal(pg[k], pe[k])
            np.testing.assert_array_equal(pg[k], out[0][True][1][k])


def _fallback_worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    try:
        from cbfssm.hip import train
        w, p, (u, y), noise = _case()
        st = train.HipTrainStep(w.model_config(), {k: torch.tensor(v, device='cuda:0') for k, v in p.items()}, 'cuda:0', dist,
                                graph=True)
        if rank == 0:
            uu, yy, nz, weight = u, y, noise, 1.0              # the whole mini-batch
        else:
            # an empty shard: the one-sequence stand-in joins the collective with weight 0 (model/cbfssm.py) ...
            uu, yy, weight = u[:1], y[:1], 0.0
            nz = {k: np.ascontiguousarray(v[:, :, :1] if k != 'eps_f' else v[:, :1]) for k, v in noise.items()}
            # ... and this rank cannot capture

            class _Broken:
                def __init__(self, *a, **k):
                    raise RuntimeError('capture refused (test)')
            torch.cuda.graph = _Broken
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            loss = float(st.step(uu, yy, nz, weight=weight))
        assert st.use_graph == (rank == 0)
        out[rank] = (loss, {k: v.detach().cpu().numpy().copy() for k, v in st.params.items()})
    finally:
        dist.destroy_process_group()


def test_capture_failure_on_an_empty_shard_keeps_its_zero_weight():
    """A data-parallel rank whose HIP-graph capture fails falls back to eager launches; when that rank holds an empty shard
    (a one-sequence stand-in with weight 0) the fallback must keep the weight, or sequence 0 is counted twice."""
    from cbfssm.hip import train
    w, p, (u, y), noise = _case()
    ref = train.HipTrainStep(w.model_config(), {k: torch.tensor(v, device='cuda:0') for k, v in p.items()}, 'cuda:0')
    loss = float(ref.step(u, y, noise))
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_fallback_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    for rank in (0, 1):
        assert out[rank][0] == pytest.approx(loss, rel=1e-10)
        for k in train.PARAM_NAMES:
            np.testing.assert_allclose(out[rank][1][k], ref.params[k].cpu().numpy(), rtol=1e-9, atol=1e-12)


# ---------------------------------------------------------------------------------------------------------------------
# stash-mode tile (M > 112): the contracted K^-1-adjoint images ride in the same flat buffer -> still ONE all-reduce
# ---------------------------------------------------------------------------------------------------------------------
def _stash_case():
    from cbfssm import synthetic as syn
    w = syn.tiny(M=130, dim_x=9, dim_u=3, dim_y=2, T=13, B=4, S=6, recog_len=3)
    return w, syn.perturb_params(syn.make_params(w), scale=0.1), syn.make_inputs(w), syn.make_noise(w)


def _stash_worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    try:
        from cbfssm.hip import train
        from cbfssm.hip.dist_utils import shard_range
        w, p, (u, y), noise = _stash_case()
        lo, hi = shard_range(w.B, rank, world)
        nz = {'hid_b': noise['hid_b'][:, :, lo:hi], 'eps_b': noise['eps_b'][:, :, lo:hi],
              'eps_f': noise['eps_f'][:, lo:hi]}
        nz = {k: np.ascontiguousarray(v) for k, v in nz.items()}
        eng = train.HipElboGrad(w.model_config(), 'cuda:0', dist)
        assert eng.stash
        calls = []
        orig = dist.all_reduce

        def counting(t, *a, **k):
            calls.append(int(t.numel()))
            return orig(t, *a, **k)
        dist.all_reduce = counting
        try:
            params = {k: torch.tensor(v, device='cuda:0') for k, v in p.items()}
            loss, grads, _ = eng.loss_and_grads(params, u[lo:hi], y[lo:hi], nz)
        finally:
            dist.all_reduce = orig
        out[rank] = (float(loss), {k: g.cpu().numpy() for k, g in grads.items()}, calls, int(eng.red.numel()))
    finally:
        dist.destroy_process_group()


def test_sharded_stash_mode_step_is_one_allreduce_and_equals_global_batch():
    from cbfssm.hip import train
    w, p, (u, y), noise = _stash_case()
    eng = train.HipElboGrad(w.model_config(), 'cuda:0')
    params = {k: torch.tensor(v, device='cuda:0') for k, v in p.items()}
    loss, grads, _ = eng.loss_and_grads(params, u, y, noise)
    ref = (float(loss), {k: g.cpu().numpy() for k, g in grads.items()})
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_stash_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    for rank in (0, 1):
        l, g, calls, nred = out[rank]
        assert calls == [nred], calls                       # exactly one collective, over the whole flat buffer
        assert l == pytest.approx(ref[0], rel=1e-10)
        for k in train.PARAM_NAMES:
            np.testing.assert_allclose(g[k], ref[1][k], rtol=1e-8, atol=1e-9 * np.abs(ref[1][k]).max())


# ---------------------------------------------------------------------------------------------------------------------
# RCCL itself: backend "nccl" with one rank all-reduces the real flat float64 buffer on the device
# ---------------------------------------------------------------------------------------------------------------------
def _nccl_worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0),
                            timeout=datetime.timedelta(seconds=120))
    try:
        from cbfssm.hip import train
        w, p, (u, y), noise = _case()
        eng = train.HipElboGrad(w.model_config(), 'cuda:0', None)
        params = {k: torch.tensor(v, device='cuda:0') for k, v in p.items()}
        state = eng._grads_local(params, u, y, noise)
        before = eng.red.clone()
        dist.all_reduce(eng.red)                            # RCCL sum over the one rank: the buffer must come back as is
        torch.cuda.synchronize()
        same = bool(torch.equal(before, eng.red))
        loss, grads, _ = eng._grads_finish(state)
        out[0] = (same, float(loss), dist.get_backend(), int(eng.red.numel()), str(eng.red.dtype))
    finally:
        dist.destroy_process_group()


def test_rccl_allreduce_accepts_the_flat_f64_buffer():
    from cbfssm.hip import train
    w, p, (u, y), noise = _case()
    eng = train.HipElboGrad(w.model_config(), 'cuda:0')
    params = {k: torch.tensor(v, device='cuda:0') for k, v in p.items()}
    loss, _, _ = eng.loss_and_grads(params, u, y, noise)
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_nccl_worker, args=(1, _free_port(), out), nprocs=1, join=True)
    same, l, backend, n, dt = out[0]
    assert backend == 'nccl' and same and dt == 'torch.float64' and n > 1000
    assert l == pytest.approx(float(loss), rel=1e-12)


# ---------------------------------------------------------------------------------------------------------------------
# the drop-in flow itself under two ranks: Trainer + Outputs through cbfssm.model.CBFSSM, NO seed in the config (every
# rank draws its own initial values, as the reference's unseeded numpy initialisers would), mini-batches that leave a
# rank without a sequence
# ---------------------------------------------------------------------------------------------------------------------
def _dropin_worker(rank, world, port, root_dir, out):
    os.environ.update({'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(port), 'WORLD_SIZE': str(world),
                       'RANK': str(rank), 'LOCAL_RANK': '0', 'CBFSSM_DIST_BACKEND': 'gloo'})
    try:
        from cbfssm.datasets import make_synthetic_ds
        from cbfssm.training import Trainer
        from cbfssm.outputs import Outputs
        from cbfssm.model import CBFSSM
        ds_sel = make_synthetic_ds(dim_u=1, dim_y=1, n_train=400, n_test=160, seed=1)
        dim_x = 3
        cfg = {'ds': ds_sel, 'batch_size': 3, 'shuffle': 10000,                     # no 'seed'
               'dim_x': dim_x, 'ind_pnt_num': 20, 'samples': 10, 'learning_rate': 0.05,
               'loss_factors': np.asarray([1., 0.]), 'k_factor': 5., 'recog_len': 8,
               'zeta_pos': 2., 'zeta_mean': 0.05 ** 2, 'zeta_var': 0.01 ** 2,
               'var_x': np.asarray([0.002 ** 2] * dim_x), 'var_y': np.asarray([1. ** 2] * dim_x),
               'gp_var': 0.5 ** 2, 'gp_len': 2.}
        ds = ds_sel(40, 20)                    # 19 training windows: six mini-batches of 3 and a last one of 1
        assert ds.train_in_batch.shape[0] % 3 == 1
        model = CBFSSM(cfg)
        own_draw = model._init_values['f.zeta_pos'].copy()
        outputs = Outputs(root_dir)
        outputs.set_ds(ds)
        outputs.set_model(model, root_dir)
        trainer = Trainer(model, root_dir)
        trainer.train(ds, 2)
        outputs.set_trainer(trainer)
        outputs.create_all()
        out[rank] = (own_draw, model._opt.flat.cpu().numpy(), list(trainer.train_all), list(trainer.test_all),
                     outputs.get_last_rmse())
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_dropin_flow_two_ranks_unseeded(tmp_path):
    root_dir = str(tmp_path / 'exp')
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_dropin_worker, args=(2, _free_port(), root_dir, out), nprocs=2, join=True)
    d0, p0, tr0, te0, rm0 = out[0]
    d1, p1, tr1, te1, rm1 = out[1]
    assert not np.array_equal(d0, d1)                       # the ranks did draw different initial values ...
    np.testing.assert_array_equal(p0, p1)                   # ... and still hold identical parameters after training
    assert tr0 == tr1 and te0 == te1 and rm0 == rm1
    assert all(np.isfinite(tr0)) and np.isfinite(rm0)
    for f in ('best.ckpt', 'model.ckpt', 'mse.txt', 'var_dump.txt', 'predict_train.mat', 'predict_test.mat'):
        assert os.path.isfile(os.path.join(root_dir, f)), f
