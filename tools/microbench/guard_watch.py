"""GPU box: does the clock's guard move in a plain steady-state loop?  One full-frame engine, episode phases spread out (or in step), the host
waiting for every step (so that every sample of the guard is read): the tuner's state and the wall time of every 2 000 steps.
python tools/microbench/guard_watch.py [T] [sync|desync]"""
import sys, time
import numpy as np
import torch
sys.path.insert(0, '.')
from gym_craftingworld_amd import CraftingWorldVecEnv
T = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
desync = len(sys.argv) > 2 and sys.argv[2] == 'desync'
N = 65536
env = CraftingWorldVecEnv(N, obs_mode='pixels', size=(21, 21), max_steps=300, seed=2024)
env.reset()
if desync:
    env.set_state(step_num=((np.arange(N) * 7) % 300).astype(np.int32))
g = torch.Generator(device='cuda').manual_seed(5)
acts = torch.randint(0, 6, (512, N), device='cuda', dtype=torch.uint8, generator=g)
torch.cuda.synchronize(); t0 = time.perf_counter()
for t in range(T):
    env.step_async(acts[t % 512]); env.step_wait()
    torch.cuda.synchronize()
    if t % 2000 == 1999:
        t1 = time.perf_counter()
        print('step %5d: %.4f ms per step (host waiting for each), tuner %s' % (t + 1, (t1 - t0) / 2000 * 1e3, env.tuner_state()), flush=True)
        t0 = t1
