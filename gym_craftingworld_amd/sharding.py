"""Multi-GPU sharding of the env batch: one process per GPU, contiguous env ranges, no data-path
collective (envs never interact: ray.py holds no cross-env state, every env has its own
np_random).  The only collectives are off the hot path: a timing max and counter sums."""
import torch
import torch.distributed as dist


def shard_range(rank, world, total_envs):
    """Contiguous range [lo, hi) of global env ids owned by `rank` (remainder spread over the first ranks)."""
    if not 0 <= rank < world:
        raise ValueError('rank %d outside world %d' % (rank, world))
    base, rem = divmod(int(total_envs), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def env_seeds(seed_base, lo, hi):
    """Seed of global env e is seed_base + e, whatever rank owns it (gym.vector convention)."""
    return [int(seed_base) + e for e in range(lo, hi)]


def max_over_ranks(value, device='cpu', group=None):
    """MAX-reduce a python float over a process group (default: the default one; identity without one)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


def gather_over_ranks(value, device='cpu', group=None):
    """Every rank's python float, in rank order, on every rank (one all_gather; [value] without a process group)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [float(value)]
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t, group=group)
    return [float(o.item()) for o in out]


def sum_over_ranks(values, device='cpu'):
    """SUM-reduce a list of ints (e.g. the engine counters) over the default process group."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [int(v) for v in values]
    t = torch.tensor([int(v) for v in values], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [int(v) for v in t.tolist()]
