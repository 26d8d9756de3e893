"""bench.py's plain `--gpus N` start: the parent spawns N fresh ranks before anything touches HIP (launch.py).
CPU only: the children here are tiny Python programs (and, once, bench.py itself failing for want of a GPU)."""
import json
import os
import subprocess
import sys
import textwrap
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from gym_craftingworld_amd import launch  # noqa: E402


def test_rank_environments_match_torchrun_conventions():
    envs = launch.rank_environments(4, base_env={'PATH': '/bin', 'WORLD_SIZE': 'stale'}, port=29511)
    assert [e['RANK'] for e in envs] == ['0', '1', '2', '3']
    assert [e['LOCAL_RANK'] for e in envs] == ['0', '1', '2', '3']
    assert all(e['WORLD_SIZE'] == '4' and e['LOCAL_WORLD_SIZE'] == '4' for e in envs)
    assert all(e['MASTER_ADDR'] == '127.0.0.1' and e['MASTER_PORT'] == '29511' for e in envs)
    assert all(e['HSA_ENABLE_IPC_MODE_LEGACY'] == '0' and e['PATH'] == '/bin' for e in envs)
    # a base env that already carries the IPC setting keeps its own
    assert launch.rank_environments(1, base_env={'HSA_ENABLE_IPC_MODE_LEGACY': '1'}, port=1)[0]['HSA_ENABLE_IPC_MODE_LEGACY'] == '1'
    with pytest.raises(ValueError):
        launch.rank_environments(0)
    # one free port for all ranks of a launch, a real one
    envs = launch.rank_environments(2, base_env={})
    assert envs[0]['MASTER_PORT'] == envs[1]['MASTER_PORT'] and 1024 < int(envs[0]['MASTER_PORT']) < 65536


def test_needs_self_launch():
    assert launch.needs_self_launch(2, environ={})
    assert launch.needs_self_launch(8, environ={'RANK': '0'})
    assert not launch.needs_self_launch(1, environ={})
    assert not launch.needs_self_launch(2, environ={'WORLD_SIZE': '2'})      # torch.distributed.run set the ranks up


def _script(tmp_path, body):
    p = tmp_path / 'child.py'
    p.write_text(textwrap.dedent(body))
    return [sys.executable, str(p)]


def test_spawn_relays_rank0_stdout_only_and_waits_for_all(tmp_path):
    argv = _script(tmp_path, '''
        import json, os, time
        r = int(os.environ['RANK'])
        time.sleep(0.2 * r)
        open(os.path.join(os.path.dirname(__file__), 'rank%d.json' % r), 'w').write(json.dumps(
            {k: os.environ[k] for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}))
        print('line from rank %d' % r)
    ''')
    with open(tmp_path / 'out.txt', 'w') as f:
        rc = launch.spawn_ranks(argv, 3, stdout=f)
    assert rc == 0
    assert (tmp_path / 'out.txt').read_text() == 'line from rank 0\n'
    seen = [json.loads((tmp_path / ('rank%d.json' % r)).read_text()) for r in range(3)]
    assert [s['RANK'] for s in seen] == ['0', '1', '2'] and all(s['WORLD_SIZE'] == '3' for s in seen)
    assert len({s['MASTER_PORT'] for s in seen}) == 1


def test_spawn_returns_failing_rank_code_and_stops_the_others(tmp_path):
    argv = _script(tmp_path, '''
        import os, sys, time
        if os.environ['RANK'] == '1':
            sys.exit(7)
        time.sleep(60)
    ''')
    t0 = time.monotonic()
    rc = launch.spawn_ranks(argv, 2, stdout=subprocess.DEVNULL)
    assert rc == 7
    assert time.monotonic() - t0 < 30          # rank 0 was not waited for: it was terminated


def test_spawn_timeout(tmp_path):
    argv = _script(tmp_path, 'import time; time.sleep(60)\n')
    t0 = time.monotonic()
    assert launch.spawn_ranks(argv, 2, timeout=1.0, stdout=subprocess.DEVNULL) == 124
    assert time.monotonic() - t0 < 30


def test_bench_plain_multi_gpu_start_spawns_ranks_instead_of_refusing():
    """`python bench.py --gpus 2` (no torchrun): in this GPU-less container the child ranks must be the ones that stop
    (no GPU for them), not the parent with "needs torch.distributed.run" as in round 1."""
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
                        '--dist-backend', 'gloo'], env=env, capture_output=True, text=True, timeout=300)
    import torch
    if torch.cuda.device_count() >= 2:
        assert p.returncode == 0 and json.loads(p.stdout.strip().splitlines()[-1])['n_gpus'] == 2
    else:
        assert p.returncode != 0
        assert 'torch.distributed.run' not in p.stderr
        assert 'needs GPU' in p.stderr or 'no MI355X' in p.stderr or 'device' in p.stderr.lower()


def test_eight_ranks_start_rendezvous_over_gloo_and_relay_one_line(tmp_path):
    """The shape of the driver's N=8 scaling run, without GPUs: eight fresh ranks from spawn_ranks() rendezvous over gloo on 127.0.0.1,
    take the timing barrier, MAX-reduce and gather their times the way bench.py does, and rank 0 alone writes the line the parent relays."""
    argv = _script(tmp_path, '''
        import datetime, json, os, sys, time
        sys.path.insert(0, %r)
        import torch.distributed as dist
        from gym_craftingworld_amd.sharding import gather_over_ranks, max_over_ranks, shard_range
        dist.init_process_group('gloo', timeout=datetime.timedelta(seconds=120))
        r, w = dist.get_rank(), dist.get_world_size()
        lo, hi = shard_range(r, w, 8 * 1000 + 3)
        dist.barrier()
        mine = 0.001 * (r + 1)
        worst = max_over_ranks(mine)
        everyone = gather_over_ranks(mine)
        dist.barrier()
        dist.destroy_process_group()
        if r == 0:
            print(json.dumps(dict(n_gpus=w, worst=worst, per_rank=everyone, lo=lo, hi=hi)))
    ''' % ROOT)
    with open(tmp_path / 'out.txt', 'w') as f:
        rc = launch.spawn_ranks(argv, 8, stdout=f, timeout=240)
    assert rc == 0
    lines = (tmp_path / 'out.txt').read_text().strip().splitlines()
    d = json.loads(lines[-1])
    assert d['n_gpus'] == 8 and abs(d['worst'] - 0.008) < 1e-12 and d['per_rank'] == [0.001 * (i + 1) for i in range(8)]
    assert (d['lo'], d['hi']) == (0, 1001)


def test_bench_places_a_short_timed_region_between_all_env_time_outs():
    """bench.py's untimed warm-up is lengthened so that a timed region shorter than an episode never straddles a multiple of max_steps
    (the step on which every env of the synchronized benchmark times out: 4x a plain step); longer regions are left alone."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('cw_bench', os.path.join(ROOT, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for prewarm, W, K in ((320, 5, 20), (288, 5, 20), (320, 0, 100), (320, 10, 280), (320, 290, 20), (320, 275, 20), (0, 295, 10), (1000, 3, 299)):
        p = bench.place_short_region(prewarm, W, K, 300)
        first, last = p + W + 1, p + W + K
        assert p >= prewarm and not any(first <= m <= last for m in range(300, last + 1, 300)), (prewarm, W, K, p)
    assert bench.place_short_region(320, 5, 20, 300) == 320 and bench.place_short_region(288, 5, 20, 300) > 288
    assert bench.place_short_region(320, 20, 600, 300) == 320 and bench.place_short_region(320, 20, 300, 300) == 320


def _fake_sysfs(root, gpus, nodes):
    """gpus: [(card name, pci address, vendor, numa_node)], nodes: {node: cpulist text}"""
    for card, pci, vendor, numa in gpus:
        real = root / 'devices' / 'pci0000:00' / pci
        real.mkdir(parents=True)
        (real / 'vendor').write_text(vendor + '\n')
        (real / 'numa_node').write_text('%d\n' % numa)
        d = root / 'class' / 'drm' / card
        d.mkdir(parents=True)
        os.symlink(str(real), str(d / 'device'))
    for n, cpus in nodes.items():
        nd = root / 'devices' / 'system' / 'node' / ('node%d' % n)
        nd.mkdir(parents=True)
        (nd / 'cpulist').write_text(cpus + '\n')


def test_ranks_bind_to_the_numa_node_of_their_gpu(tmp_path):
    """bench.py --gpus N: rank r, before it touches the GPU, keeps itself on the CPUs of the NUMA node GPU r hangs off (launch.bind_to_gpu_numa), read
    from /sys/class/drm/card*/device/numa_node in PCI order -- and leaves its affinity alone whenever that cannot be known."""
    # eight GPUs on two sockets, card numbers NOT in PCI order, one non-AMD device in between
    gpus = [('card%d' % (7 - i), '0000:%02x:00.0' % (0x05 + 0x10 * i), '0x1002', 0 if i < 4 else 1) for i in range(8)]
    gpus.append(('card8', '0000:03:00.0', '0x1a03', 0))                     # (the BMC's VGA device)
    _fake_sysfs(tmp_path, gpus, {0: '0-3,8-11', 1: '4-7,12-15'})
    assert launch.gpu_numa_nodes(str(tmp_path)) == [0, 0, 0, 0, 1, 1, 1, 1]
    assert launch.node_cpus(1, str(tmp_path)) == {4, 5, 6, 7, 12, 13, 14, 15}
    seen = {}
    kw = dict(sys_root=str(tmp_path), setaffinity=lambda pid, cpus: seen.update(pid=pid, cpus=set(cpus)), getaffinity=lambda pid: set(range(16)))
    assert launch.bind_to_gpu_numa(5, environ={}, **kw) == {'node': 1, 'cpus': 8} and seen == {'pid': 0, 'cpus': {4, 5, 6, 7, 12, 13, 14, 15}}
    # a cgroup cpuset that covers half of the node: the intersection
    kw['getaffinity'] = lambda pid: {0, 1, 2, 3, 4, 5}
    assert launch.bind_to_gpu_numa(2, environ={}, **kw) == {'node': 0, 'cpus': 4} and seen['cpus'] == {0, 1, 2, 3}
    # ... and everything that must leave the affinity alone
    seen.clear()
    assert 'CW_NUMA_BIND' in launch.bind_to_gpu_numa(0, environ={'CW_NUMA_BIND': '0'}, **kw)['skipped']
    assert 'HIP_VISIBLE_DEVICES' in launch.bind_to_gpu_numa(0, environ={'HIP_VISIBLE_DEVICES': '3,2'}, **kw)['skipped']
    assert 'local rank 9' in launch.bind_to_gpu_numa(9, environ={}, **kw)['skipped']
    kw['getaffinity'] = lambda pid: {4, 5}
    assert 'affinity' in launch.bind_to_gpu_numa(0, environ={}, **kw)['skipped']
    one = tmp_path / 'single'
    _fake_sysfs(one, [('card0', '0000:05:00.0', '0x1002', -1)], {})
    assert 'numa_node' in launch.bind_to_gpu_numa(0, environ={}, sys_root=str(one), setaffinity=kw['setaffinity'], getaffinity=kw['getaffinity'])['skipped']
    assert launch.bind_to_gpu_numa(0, environ={}, sys_root=str(tmp_path / 'nothing'), setaffinity=kw['setaffinity'],
                                   getaffinity=kw['getaffinity'])['skipped'].startswith('sysfs lists 0')
    assert seen == {}
    # the real host, whatever it is: never raises, never binds to an empty set
    r = launch.bind_to_gpu_numa(0, environ={'CW_NUMA_BIND': '0'})
    assert 'skipped' in r
