"""GPU box: per-step latency of the N=1 gym.Env-shaped facade (numpy in/out, no auto-reset), the reference's own loop."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import gym_craftingworld_amd as cw

for env_id in ('craftingworld-v3', 'craftingworldflat-v3', 'craftingworldonehot-v3'):
    env = cw.make(env_id)
    env.seed(0)
    rng = np.random.RandomState(0)
    acts = rng.randint(0, 6, size=5000)
    env.reset()
    for a in acts[:200]:
        _, _, d, _ = env.step(a)
        if d:
            env.reset()
    n_reset, t0 = 0, time.perf_counter()
    for a in acts:
        _, _, d, _ = env.step(a)
        if d:
            env.reset(); n_reset += 1
    dt = time.perf_counter() - t0
    print('%s: %.1f us per step (%d steps, %d resets inside)' % (env_id, dt / len(acts) * 1e6, len(acts), n_reset))
    env.close()
