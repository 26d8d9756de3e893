"""The C ABI used from plain C (examples/c_api_demo.c: no Python, no torch): build it with gcc,
run it on the GPU, and check every number it prints against the CPU oracle."""
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build():
    exe = os.path.join(ROOT, 'examples', 'c_api_demo')
    subprocess.check_call(['make', '-C', os.path.join(ROOT, 'gym_craftingworld_amd', 'csrc')], stdout=subprocess.DEVNULL)
    subprocess.check_call(['gcc', '-O2', '-std=c99', '-D__HIP_PLATFORM_AMD__', '-I', os.path.join(ROOT, 'include'),
                           '-I', '/opt/rocm/include', '-o', exe, os.path.join(ROOT, 'examples', 'c_api_demo.c'),
                           '-L', os.path.join(ROOT, 'gym_craftingworld_amd'), '-lcraftingworld', '-L', '/opt/rocm/lib',
                           '-lamdhip64', '-Wl,-rpath,' + os.path.join(ROOT, 'gym_craftingworld_amd'), '-Wl,-rpath,/opt/rocm/lib'])
    return exe


def test_c_demo_builds_as_plain_c():
    assert os.path.exists(_build())


@pytest.mark.gpu
def test_c_demo_matches_oracle():
    from oracle import OracleBatch
    out = subprocess.check_output([_build()], timeout=300).decode()
    m = re.search(r'env_steps (\d+) episodes (\d+) successes (\d+) reward_sum_sampled (-?\d+) frame17_fnv ([0-9a-f]{8})', out)
    assert m, out
    N, T = 1024, 600
    states = [np.random.RandomState(5000 + i).get_state() for i in range(N)]
    ora = OracleBatch(N, rng_states=[(s[1], s[2]) for s in states], size=(21, 21), max_steps=300)
    ora.reset()
    s = np.uint32(12345)
    acts = np.empty(T * N, dtype=np.int8)
    x = 12345
    for i in range(T * N):                       # the demo's LCG
        x = (x * 1664525 + 1013904223) & 0xFFFFFFFF
        acts[i] = (x >> 8) % 6
    acts = acts.reshape(T, N)
    total, rew, don = ora.rollout(acts, nthreads=8, record=True)
    sampled = sum(int(rew[t].sum()) for t in range(T) if t % 100 == 99 or t == T - 1)
    frame = ora.envs[17].state()['obs'].reshape(-1)
    fnv = 2166136261
    for b in frame.tolist():
        fnv = ((fnv ^ b) * 16777619) & 0xFFFFFFFF
    assert int(m.group(1)) == N * T
    assert int(m.group(2)) == int(don.sum())
    assert int(m.group(3)) == int((rew == 300).sum())
    assert int(m.group(4)) == sampled
    assert m.group(5) == '%08x' % fnv
    # second line: the single-env loop through cw_step_resident (no kernel launch per step), against the oracle's single env
    from oracle import OracleEnv
    m1 = re.search(r'single_env steps (\d+) episodes (\d+) reward_sum (-?\d+) achieved_trace ([0-9a-f]{8}) frame_fnv ([0-9a-f]{8}) us_per_step_incl_resets ([0-9.]+)', out)
    assert m1, out
    st = np.random.RandomState(7777).get_state()
    env = OracleEnv(rng_state=(st[1], st[2]), size=(21, 21), max_steps=120)
    obs = env.reset()
    x, rsum, episodes, trace = 99, 0, 0, 2166136261
    for t in range(3000):
        x = (x * 1664525 + 1013904223) & 0xFFFFFFFF
        obs, r, d, info = env.step((x >> 8) % 6)
        rsum += r
        ach = sum(int(b) << i for i, b in enumerate(np.asarray(info['achieved_goal']).reshape(-1)))
        trace = ((trace ^ ach) * 16777619) & 0xFFFFFFFF
        if d:
            episodes += 1
            obs = env.reset()
    f1 = 2166136261
    for b in np.asarray(obs['observation'], dtype=np.uint8).reshape(-1).tolist():
        f1 = ((f1 ^ b) * 16777619) & 0xFFFFFFFF
    assert (int(m1.group(1)), int(m1.group(2)), int(m1.group(3))) == (3000, episodes, rsum)
    assert m1.group(4) == '%08x' % trace and m1.group(5) == '%08x' % f1
    assert 0 < float(m1.group(6)) < 200.0                # (measured 4.6 us; the launch path and the reference's own step are ~17 us -- a sanity bound, not a perf floor: box load)
