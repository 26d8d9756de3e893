"""Build-container-only: SURVEY §8(d)(iii) calibration.  Times the reference's own Python step()/reset()
loop and the C port (oracle/, one thread) on the same workload -- 21x21, max_steps=300, uniform random
actions, reset after every done -- and writes the ratio to tests/golden/cpu_calibration.json so that a
`cpu_baseline` of kind "port" measured on the GPU box can be related to the true reference, which
cannot travel.  Also checks that both sides produced the same rewards on the timed sample.

    python tools/calibrate_cpu.py [--seconds 10]
"""
import argparse
import json
import os
import platform
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

from refharness import import_reference, make_ref_env  # noqa: E402


def time_reference(cls, seconds, size, max_steps, seed, acts):
    env = make_ref_env(cls, np.random.RandomState(seed), size=(size, size), max_steps=max_steps)
    env.reset()
    n, rews, t0 = 0, [], time.perf_counter()
    while True:
        _, r, d, _ = env.step(int(acts[n % len(acts)]))
        rews.append(r)
        n += 1
        if d:
            env.reset()
        if (n & 1023) == 0 and time.perf_counter() - t0 >= seconds:
            break
    dt = time.perf_counter() - t0
    return n, np.asarray(rews, np.int32), dt


def time_port(n_steps, size, max_steps, seed, acts):
    from oracle import OracleBatch
    st = np.random.RandomState(seed).get_state()
    b = OracleBatch(1, rng_states=[(st[1], st[2])], size=(size, size), max_steps=max_steps)
    b.reset()
    a = np.resize(acts, n_steps).astype(np.int8).reshape(n_steps, 1)
    t0 = time.perf_counter()
    _, rew, _ = b.rollout(a, nthreads=1, record=True)
    dt = time.perf_counter() - t0
    return rew[:, 0], dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--seconds', type=float, default=10.0)
    ap.add_argument('--size', type=int, default=21)
    ap.add_argument('--max-steps', type=int, default=300)
    ap.add_argument('--out', default=os.path.join(os.path.dirname(HERE), 'tests', 'golden', 'cpu_calibration.json'))
    a = ap.parse_args()
    classes = import_reference()
    acts = np.random.RandomState(1).randint(0, 6, size=1 << 16)
    n, rew_ref, dt_ref = time_reference(classes['ray'], a.seconds, a.size, a.max_steps, 4242, acts)
    # the port needs far more steps for a stable timing: repeat the action tape
    n_port = max(n, 64) * 64
    rew_port, dt_port = time_port(n_port, a.size, a.max_steps, 4242, acts)
    assert (rew_port[:n] == rew_ref).all(), 'port and reference disagree on the timed sample'
    ref_rate, port_rate = n / dt_ref, n_port / dt_port
    out = dict(workload='%dx%d, max_steps=%d, uniform random actions, reset after every done, 1 env, 1 core'
               % (a.size, a.size, a.max_steps),
               reference_python_env_steps_per_s=ref_rate, reference_steps_timed=n,
               port_c_env_steps_per_s_1thread=port_rate, port_steps_timed=n_port,
               port_over_reference=port_rate / ref_rate, rewards_equal_on_reference_sample=True,
               host='%s, %d CPUs visible' % (platform.processor() or platform.machine(), os.cpu_count()),
               numpy=np.__version__, python=platform.python_version(),
               note='measured in the build container by tools/calibrate_cpu.py; the reference is imported '
                    'read-only from /root/reference under tools/gym_stub')
    with open(a.out, 'w') as f:
        json.dump(out, f, indent=1)
        f.write('\n')
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
