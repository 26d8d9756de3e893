"""N>1 path on CPU: world_size-2 gloo processes shard an env batch the way bench.py / a multi-GPU
user does (contiguous ranges, global-env-id seeds, no data-path collective) and the concatenation
of the shards must equal the single-process batch bit for bit.  The per-shard compute is the CPU
oracle here (test infrastructure; the HIP engine needs a GPU -- its shard equivalence is
tests/test_hip_parity.py::test_shard_equivalence_and_batch_position_invariance)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

N_TOTAL, T = 37, 60          # odd on purpose: uneven shards
KW = dict(size=(6, 6), max_steps=20)


def _run_shard(lo, hi, acts):
    from oracle import OracleBatch
    sts = [np.random.RandomState(1000 + e).get_state() for e in range(lo, hi)]
    b = OracleBatch(hi - lo, rng_states=[(s[1], s[2]) for s in sts], **KW)
    b.reset()
    total, rew, don = b.rollout(acts[:, lo:hi].astype(np.int8), nthreads=1, record=True)
    grids = np.stack([s['grid'] for s in b.states()])
    return total, rew, don, grids


def _worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from gym_craftingworld_amd.sharding import env_seeds, max_over_ranks, shard_range, sum_over_ranks
    lo, hi = shard_range(rank, world, N_TOTAL)
    assert env_seeds(1000, lo, hi) == list(range(1000 + lo, 1000 + hi))
    acts = np.random.RandomState(3).randint(0, 6, size=(T, N_TOTAL))
    total, rew, don, grids = _run_shard(lo, hi, acts)
    dist.barrier()
    elapsed = max_over_ranks(0.5 + rank)            # the slowest rank defines the step time
    steps, = sum_over_ranks([total])
    q.put((rank, lo, hi, rew, don, grids, elapsed, steps))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shards_equal_single_batch():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted([q.get(timeout=120) for _ in procs], key=lambda x: x[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert (out[0][1], out[0][2], out[1][1], out[1][2]) == (0, 19, 19, 37)
    acts = np.random.RandomState(3).randint(0, 6, size=(T, N_TOTAL))
    total, rew, don, grids = _run_shard(0, N_TOTAL, acts)
    assert np.array_equal(rew, np.concatenate([o[3] for o in out], axis=1))
    assert np.array_equal(don, np.concatenate([o[4] for o in out], axis=1))
    assert np.array_equal(grids, np.concatenate([o[5] for o in out], axis=0))
    assert all(o[6] == 1.5 for o in out)            # max over ranks seen by every rank
    assert all(o[7] == total == N_TOTAL * T for o in out)


def test_shard_range_covers_everything():
    from gym_craftingworld_amd.sharding import shard_range
    for world in (1, 2, 3, 4, 8):
        for total in (8, 65536, 1048576, 1000003):
            r = [shard_range(g, world, total) for g in range(world)]
            assert r[0][0] == 0 and r[-1][1] == total
            assert all(r[i][1] == r[i + 1][0] for i in range(world - 1))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1
    with pytest.raises(ValueError):
        shard_range(2, 2, 10)


def _rccl_worker(rank, world, port, q, failing_rank, fail_at):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import time
    from gym_craftingworld_amd.sharding import agree_on_rccl

    # stands in for dist.new_group(backend='nccl') -- there is no GPU here -- with RCCL's semantics: creating the group sends nothing (the
    # communicator is built lazily, by the group's first collective); collectives "on" it go over the default gloo group
    seen_at_create = []

    def new_group(backend, timeout):
        seen_at_create.append(os.environ.get('TORCH_NCCL_BLOCKING_WAIT'))      # (what torch reads when it creates the group)
        if rank == failing_rank and fail_at == 'create':
            raise RuntimeError('injected: ncclCommInitRank failed')
        return 'fake-rccl'

    real_all_reduce = dist.all_reduce

    def all_reduce(t, op=dist.ReduceOp.SUM, group=None):
        if group == 'fake-rccl':
            if fail_at == 'probe':
                if rank == failing_rank:
                    raise RuntimeError('injected: ncclAllReduce failed')
                time.sleep(2.0)                       # the healthy ranks wait for a peer that never arrives: blocking wait, then the group's timeout
                raise RuntimeError('injected: timed out waiting for the probe all-reduce (blocking wait)')
            group = None
        return real_all_reduce(t, op=op, group=group)

    dist.all_reduce = all_reduce
    t0 = time.time()
    os.environ.pop('TORCH_NCCL_BLOCKING_WAIT', None)
    if rank % 2:
        os.environ['TORCH_NCCL_BLOCKING_WAIT'] = '0'              # (a caller who set it: kept for the group, and still there afterwards)
    group, why = agree_on_rccl('cpu', timeout_s=20, new_group=new_group)
    # library code: the variable is in force while THIS group is created and the process's environment is as the caller left it afterwards
    assert seen_at_create == ['0' if rank % 2 else '1'], seen_at_create
    assert os.environ.get('TORCH_NCCL_BLOCKING_WAIT') == ('0' if rank % 2 else None)
    q.put((rank, group is not None, why, time.time() - t0))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world,failing_rank,fail_at,expect_group', [
    (2, 1, 'create', False), (2, 0, 'create', False), (2, 1, 'probe', False), (2, None, None, True),
    (8, 3, 'create', False), (8, 5, 'probe', False), (8, None, None, True)])
def test_rccl_agreement_when_one_rank_fails(world, failing_rank, fail_at, expect_group):
    """bench.py's timing collectives go over RCCL only if it works on EVERY rank (sharding.agree_on_rccl).  One rank of two -- or a MIDDLE rank of the
    eight of the driver's scaling run -- fails (injected; the group is a gloo one here), either when it creates its RCCL group or in the probe
    all-reduce (its peers then wait in theirs until the blocking wait times out): every rank must land on gloo, within seconds of one another and
    without waiting out the 20-s group timeout at the create stage.  Nobody fails: everybody keeps the group."""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_rccl_worker, args=(r, world, port, q, failing_rank, fail_at)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted([q.get(timeout=180) for _ in procs], key=lambda x: x[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert [o[1] for o in out] == [expect_group] * world, out
    assert all(o[3] < 15.0 for o in out), out                  # (nobody waited out the 20-s group timeout)
    if not expect_group:
        assert 'injected' in out[failing_rank][2]
        for r in range(world):
            if r != failing_rank:
                assert out[r][2] == ('on another rank' if fail_at == 'create' else out[r][2]) and (fail_at == 'create' or 'timed out' in out[r][2]), out[r]
