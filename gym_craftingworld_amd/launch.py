"""One process per GPU without torchrun: `python bench.py --gpus N` started plainly becomes a parent that starts N fresh
child ranks and relays rank 0's output.  The parent never touches HIP (no torch.cuda / engine call before or after the
spawn), so no process that has initialised the GPU is ever replaced or forked: every rank is a new interpreter.

Envs never interact (ray.py holds no cross-env state), so ranks own contiguous env ranges (sharding.py) and the only
collectives are the timing barrier and the max-over-ranks in bench.py; this module is only process plumbing.
"""
import os
import socket
import subprocess
import sys
import time


def free_port(addr='127.0.0.1'):
    """A TCP port that was free a moment ago on `addr` (the rendezvous address is always 127.0.0.1: one node)."""
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind((addr, 0))
        return s.getsockname()[1]


def rank_environments(n_ranks, base_env=None, port=None, addr='127.0.0.1'):
    """The environment of each of `n_ranks` local ranks: RANK / LOCAL_RANK / WORLD_SIZE / LOCAL_WORLD_SIZE / MASTER_ADDR /
    MASTER_PORT exactly as torch.distributed.run sets them, on top of `base_env` (default: this process's)."""
    if n_ranks < 1:
        raise ValueError('n_ranks must be >= 1')
    base = dict(os.environ if base_env is None else base_env)
    port = free_port(addr) if port is None else int(port)
    envs = []
    for r in range(n_ranks):
        e = dict(base)
        e.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                 MASTER_ADDR=addr, MASTER_PORT=str(port), CW_BENCH_CHILD='1')
        e.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC only on this pool (RCCL needs it)
        envs.append(e)
    return envs


def needs_self_launch(n_gpus, environ=None):
    """True when `--gpus n_gpus` was asked for but no launcher set up the ranks (WORLD_SIZE absent)."""
    environ = os.environ if environ is None else environ
    return n_gpus > 1 and 'WORLD_SIZE' not in environ


def spawn_ranks(argv, n_ranks, timeout=None, poll=0.05, stdout=None):
    """Start `argv` once per rank, wait for all, return the worst exit code.  Rank 0's stdout goes to `stdout` (default:
    ours) unchanged; the other ranks' stdout is dropped, everyone's stderr is ours.  If a rank fails, or `timeout` seconds
    pass, the remaining ranks (exactly the PIDs started here) are terminated, then killed."""
    envs = rank_environments(n_ranks)
    procs = []
    try:
        for r, e in enumerate(envs):
            out = (stdout if stdout is not None else None) if r == 0 else subprocess.DEVNULL
            procs.append(subprocess.Popen(argv, env=e, stdout=out, stdin=subprocess.DEVNULL))
        t0 = time.monotonic()
        worst = 0
        alive = list(procs)
        while alive:
            for p in list(alive):
                rc = p.poll()
                if rc is None:
                    continue
                alive.remove(p)
                if rc != 0:
                    worst = rc if worst == 0 else worst
            if worst != 0 or (timeout is not None and time.monotonic() - t0 > timeout):
                if worst == 0:
                    worst = 124
                break
            time.sleep(poll)
        return worst
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        deadline = time.monotonic() + 10.0
        for p in procs:
            while p.poll() is None and time.monotonic() < deadline:
                time.sleep(0.05)
            if p.poll() is None:
                p.kill()
                p.wait()


def self_launch(n_ranks, script, args, timeout=None):
    """Re-run `script args` as n_ranks fresh ranks and return the exit code for the parent to exit with."""
    return spawn_ranks([sys.executable, script] + list(args), n_ranks, timeout=timeout)
