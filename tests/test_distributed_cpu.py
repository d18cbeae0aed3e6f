"""world_size-2 gloo tests (CPU) of the N>1 host path: shard arithmetic, the flat all-reduce helper, identical
iteration order on every rank.  The sharded-vs-global equality of the kernels themselves is test_distributed_gpu.py."""
import datetime
import os
import socket
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cbfssm.hip.dist_utils import shard_range, all_reduce_sum, broadcast_seed
from cbfssm.model.base_model import BaseModel


def test_shard_range_partitions():
    for n in (1, 5, 16, 255, 256, 2048):
        for world in (1, 2, 3, 8):
            parts = [shard_range(n, r, world) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in parts]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    try:
        # one flat float64 buffer per step: [slab_f | slab_b | scalars]
        t = torch.arange(1000, dtype=torch.float64) * (rank + 1)
        all_reduce_sum(t, dist)
        ok1 = bool(torch.equal(t, torch.arange(1000, dtype=torch.float64) * sum(range(1, world + 1))))
        seed = broadcast_seed(100 + rank, dist)
        m = BaseModel({'batch_size': 6, 'shuffle': 50, 'seed': seed})
        a = np.arange(20 * 2 * 1, dtype=float).reshape(20, 2, 1)
        m.load_ds(None, a, a)
        first = m._next_batch()[0][:, 0, 0]
        lo, hi = shard_range(first.shape[0], rank, world)
        mine = torch.tensor(first[lo:hi])
        gathered = [torch.zeros(3, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(gathered, mine)
        ok2 = bool(np.array_equal(torch.cat(gathered).numpy(), first))
        out[rank] = (ok1, seed, ok2)
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_allreduce_and_identical_iteration_order():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert out[0][0] and out[1][0]
    assert out[0][1] == out[1][1] == 100          # rank 0's seed wins
    assert out[0][2] and out[1][2]


def _worker_params(rank, world, port, out, tmp):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    try:
        from cbfssm.hip.dist_utils import broadcast_tensor, is_writer, barrier, active
        # every rank draws its own (unseeded) initial values; rank 0's win
        flat = torch.full((1000,), float(rank + 1), dtype=torch.float64) + torch.rand(1000, dtype=torch.float64)
        mine = flat.clone()
        broadcast_tensor(flat, dist, src=0)
        gathered = [torch.zeros(1000, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(gathered, flat)
        same = all(torch.equal(gathered[0], g) for g in gathered)
        # a rank with an empty shard joins the one all-reduce with weight 0
        red = torch.arange(8, dtype=torch.float64) * (rank + 1)
        weight = 1.0 if rank == 0 else 0.0
        red.mul_(weight)
        all_reduce_sum(red, dist)
        ok_w = bool(torch.equal(red, torch.arange(8, dtype=torch.float64)))
        # rank 0 writes, the others wait for the file
        path = os.path.join(tmp, 'ckpt')
        if is_writer():
            with open(path, 'w') as f:
                f.write('x')
        barrier()
        out[rank] = (same, bool(torch.equal(flat, mine)) == (rank == 0), ok_w, is_writer() == (rank == 0),
                     os.path.exists(path), active() is not None)
    finally:
        dist.destroy_process_group()


def test_two_rank_parameter_broadcast_zero_weight_shard_and_single_writer(tmp_path):
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_params, args=(world, _free_port(), out, str(tmp_path)), nprocs=world, join=True)
    for rank in range(world):
        assert all(out[rank]), (rank, out[rank])
