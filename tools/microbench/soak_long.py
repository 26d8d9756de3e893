"""GPU box: a long soak of the final build -- T steps of 65 536 full-frame envs with the episode phases spread out against the dirty-cell engine:
all three frame arrays every 2 000 steps, results / counters / random streams at the end.  python tools/microbench/soak_long.py [T]"""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from gym_craftingworld_amd import CraftingWorldVecEnv
T = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
N = 65536
kw = dict(size=(21, 21), max_steps=300, seed=2024)
full = CraftingWorldVecEnv(N, obs_mode='pixels', **kw)
dirty = CraftingWorldVecEnv(N, obs_mode='pixels_dirty', **kw)
full.reset(); dirty.reset()
phase = ((np.arange(N) * 7) % 300).astype(np.int32)
full.set_state(step_num=phase); dirty.set_state(step_num=phase)
g = torch.Generator(device='cuda').manual_seed(5)
acts = torch.randint(0, 6, (512, N), device='cuda', dtype=torch.uint8, generator=g)
bad = 0
for t in range(T):
    a = acts[t % 512]
    full.step_async(a); full.step_wait(); dirty.step_async(a); dirty.step_wait()
    if t % 2000 == 1999 or t == T - 1:
        for k in ('observation', 'desired_goal', 'init_observation'):
            if not torch.equal(full._observation()[k], dirty._observation()[k]):
                bad += 1; print('step', t, k, 'DIFFERS')
        print('step %d ok, finished episodes %d, tuner %s' % (t + 1, int(full.counters[1]), full.tuner_state()), flush=True)
kf, pf = full.get_rng_states(); kd, pd = dirty.get_rng_states()
same = np.array_equal(kf, kd) and np.array_equal(pf, pd) and torch.equal(full.reward, dirty.reward) and torch.equal(full.done, dirty.done)
print('random streams and last results equal:', same, '; frame comparisons that differed:', bad)
sys.exit(0 if same and bad == 0 else 1)
