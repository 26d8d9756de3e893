"""GPU box: the all-env time-out step (every env finishes on the same step: the step kernel writes two frames per env, then the sweep) --
per-step wall time over two episode lengths, the slowest steps listed.  python tools/microbench/storm_step.py [envs] [size]"""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from gym_craftingworld_amd import CraftingWorldVecEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
size = int(sys.argv[2]) if len(sys.argv) > 2 else 21
env = CraftingWorldVecEnv(n, size=(size, size), obs_mode='pixels', max_steps=100, device='cuda:0')
env.seed(0); env.reset()
g = torch.Generator(device='cuda').manual_seed(1)
acts = torch.randint(0, 6, (260, n), device='cuda', dtype=torch.uint8, generator=g)
for t in range(50):
    env.step_async(acts[t]); env.step_wait()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(211)]
torch.cuda.synchronize()
ev[0].record()
for t in range(210):
    env.step_async(acts[50 + t]); env.step_wait(); ev[t + 1].record()
torch.cuda.synchronize()
ms = np.array([ev[t].elapsed_time(ev[t + 1]) for t in range(210)])
order = np.argsort(ms)[::-1]
print('median step %.4f ms; slowest: %s' % (np.median(ms), ', '.join('step %d %.4f' % (50 + i + 1, ms[i]) for i in order[:4])))
print('the step after each time-out step: %s' % ', '.join('%.4f' % ms[i + 1] for i in order[:2] if i + 1 < 210))
