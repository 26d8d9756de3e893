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


def gather_objects_over_ranks(obj):
    """Every rank's picklable `obj`, in rank order, on every rank -- over the DEFAULT (gloo, CPU) group: the control plane, off the timed regions
    ([obj] without a process group)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [obj]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, obj)
    return out


def sum_over_ranks(values, device='cpu'):
    """SUM-reduce a list of ints (e.g. the engine counters) over the default process group."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [int(v) for v in values]
    t = torch.tensor([int(v) for v in values], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [int(v) for v in t.tolist()]


def agree_on_rccl(device, timeout_s=60, new_group=None, log=None):
    """The timing collectives of a multi-rank run go over RCCL (backend "nccl") -- a second process group beside the default gloo one -- if it works
    on EVERY rank, over gloo otherwise: either everybody uses RCCL or nobody does, whatever subset of ranks saw a failure.  Two rounds, each closed
    by a MIN-reduce of the ranks' "ok" flags over gloo BEFORE anybody depends on the other ranks having got as far: (1) creating the group (nothing is
    sent yet: a rank that fails here keeps its peers out of RCCL's first collective, where they would otherwise wait out the group's timeout);
    (2) one all-reduce on it.  A group that failed is destroyed.  -> (group or None, None or the reason as text).  Needs the default (gloo) group.
    THIS group is created with TORCH_NCCL_BLOCKING_WAIT=1 (unless the caller's environment says otherwise): a rank whose peers never arrive in the
    probe -- theirs threw -- gets an exception after `timeout_s` instead of a kernel that spins until somebody kills the job, reaches the second
    round and lands on gloo with everybody else.  The variable is read when a group is created and is put back as it was right after: this is
    library code, and the NCCL groups the host application creates later keep torch's own defaults.  For THIS group blocking wait also means
    (ProcessGroupNCCL switches asynchronous error handling off with it) that a timing collective which hangs later does not end the rank through
    the watchdog: it raises from the collective's wait() after the group's timeout -- bench.py's timing collectives are all waited on."""
    import datetime
    import os
    had = os.environ.get('TORCH_NCCL_BLOCKING_WAIT')
    os.environ.setdefault('TORCH_NCCL_BLOCKING_WAIT', '1')      # (read when the group is created)
    new_group = new_group or dist.new_group
    world = dist.get_world_size()

    def agreed(ok):
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return int(flag.item()) == 1

    group, why = None, ''
    try:
        group = new_group(backend='nccl', timeout=datetime.timedelta(seconds=timeout_s))
    except Exception as exc:  # noqa: BLE001
        why = 'creating the group: %s: %s' % (type(exc).__name__, str(exc).splitlines()[0] if str(exc) else '')
    finally:                                                    # (the process's environment as the caller left it)
        if had is None:
            os.environ.pop('TORCH_NCCL_BLOCKING_WAIT', None)
        else:
            os.environ['TORCH_NCCL_BLOCKING_WAIT'] = had
    if agreed(group is not None):
        try:
            probe = torch.ones(1, device=device)
            dist.all_reduce(probe, group=group)
            if getattr(probe, 'is_cuda', False):
                torch.cuda.synchronize(device)
            if int(probe.item()) != world:
                why = 'the probe all-reduce returned %r, not the world size' % probe.item()
        except Exception as exc:  # noqa: BLE001
            why = 'the probe all-reduce: %s: %s' % (type(exc).__name__, str(exc).splitlines()[0] if str(exc) else '')
        if agreed(not why):
            return group, None
    if why and log:
        log('RCCL group failed on rank %d (%s)' % (dist.get_rank(), why))
    if group is not None:
        try:
            dist.destroy_process_group(group)
        except Exception:  # noqa: BLE001
            pass
    return None, why or 'on another rank'
