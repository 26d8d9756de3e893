"""GPU box: per-step latency of the N=1 gym.Env-shaped facade (numpy in/out, no auto-reset), the reference's own loop --
with the resident stepper (default: doorbell + spin) and with the launch path (resident=False: kernel launch + stream sync)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import gym_craftingworld_amd as cw

for env_id, kw in (('craftingworld-v3', dict(resident=True)), ('craftingworld-v3', dict(resident=False)),
                   ('craftingworldflat-v3', dict(resident=True)), ('craftingworldflat-v3', dict(resident=False)),
                   ('craftingworldonehot-v3', dict())):
    env = cw.make(env_id, **kw)
    env.seed(0)
    rng = np.random.RandomState(0)
    acts = rng.randint(0, 6, size=5000)
    env.reset()
    for a in acts[:200]:
        _, _, d, _ = env.step(a)
        if d:
            env.reset()
    runs = []
    for rep in range(3):
        n_reset, t0 = 0, time.perf_counter()
        t_steps = 0.0
        for a in acts:
            t1 = time.perf_counter()
            _, _, d, _ = env.step(a)
            t_steps += time.perf_counter() - t1
            if d:
                env.reset(); n_reset += 1
        dt = time.perf_counter() - t0
        runs.append((dt / len(acts) * 1e6, t_steps / len(acts) * 1e6))
    print('%-24s %-18s %s us per step incl. resets | step() alone %s us  (%d steps, %d resets inside)' % (
        env_id, kw, ' '.join('%.1f' % r[0] for r in runs), ' '.join('%.1f' % r[1] for r in runs), len(acts), n_reset))
    env.close()
