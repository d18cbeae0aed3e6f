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
