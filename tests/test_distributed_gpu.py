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
