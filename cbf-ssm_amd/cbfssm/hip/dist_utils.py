"""Data-parallel helpers: sequences of a mini-batch shard over ranks as independent chains; the only exchange is
one sum all-reduce of a flat float64 buffer per step (torch.distributed: backend nccl = RCCL over xGMI on the GPUs;
gloo is used by the CPU-side tests and moves the buffer through host memory)."""
import torch


def shard_range(n, rank, world):
    """Contiguous shard [lo, hi) of n items for `rank`; sizes differ by at most one, earlier ranks get the extra."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def all_reduce_sum(t, dist):
    """In-place sum over all ranks.  `dist` is the torch.distributed module (already initialised) or None."""
    if dist is None or dist.get_world_size() == 1:
        return t
    if t.is_cuda and dist.get_backend() == 'gloo':
        h = t.detach().cpu()
        dist.all_reduce(h)
        t.copy_(h)
    else:
        dist.all_reduce(t)
    return t


def broadcast_seed(seed, dist):
    """Every rank iterates the dataset in the same order: rank 0's seed wins."""
    if dist is None or dist.get_world_size() == 1:
        return int(seed)
    t = torch.tensor([int(seed)], dtype=torch.int64)
    if dist.get_backend() != 'gloo':
        t = t.cuda()
    dist.broadcast(t, src=0)
    return int(t.item())


def broadcast_tensor(t, dist, src=0):
    """In-place broadcast of a tensor from rank `src` (parameters and optimiser state at start-up)."""
    if dist is None or dist.get_world_size() == 1:
        return t
    if t.is_cuda and dist.get_backend() == 'gloo':
        h = t.detach().cpu()
        dist.broadcast(h, src=src)
        t.copy_(h)
    else:
        dist.broadcast(t, src=src)
    return t


def active(dist_module=None):
    """torch.distributed when a process group with more than one rank is up, else None."""
    try:
        import torch.distributed as td
    except ImportError:
        return None
    if td.is_available() and td.is_initialized() and td.get_world_size() > 1:
        return td
    return None


def is_writer():
    """True on the one rank that writes checkpoints and report files (rank 0; always True without a process group)."""
    td = active()
    if td is not None:
        return td.get_rank() == 0
    import os
    return int(os.environ.get('RANK', '0')) == 0        # before the first Session of a torch.distributed.run launch


def barrier():
    td = active()
    if td is not None:
        td.barrier()
