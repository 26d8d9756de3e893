"""GPU box: the guard's PROBES -- an engine started under what the card takes (CW_TUNE_RATE_TBS, guard on) finds its way up notch by notch, keeping a faster
clock only when it pays; an engine at cw_create's own choice does not move.   CW_TUNE_VERBOSE=1 python tools/microbench/r05_guard_probe.py [start rate] [steps] [desync]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, '.')
if len(sys.argv) > 1 and float(sys.argv[1]) > 0:
    os.environ['CW_TUNE_RATE_TBS'] = sys.argv[1]
from gym_craftingworld_amd import CraftingWorldVecEnv
T = int(sys.argv[2]) if len(sys.argv) > 2 else 40000
N = 65536
env = CraftingWorldVecEnv(N, obs_mode='pixels', size=(21, 21), max_steps=300, seed=1)
env.reset()
if len(sys.argv) > 3:
    env.set_state(step_num=((np.arange(N) * 7) % 300).astype(np.int32))
acts = torch.randint(0, 6, (256, N), device='cuda', dtype=torch.uint8)
print('start', env.tuner_state(), flush=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
for t in range(T):
    env.step_async(acts[t % 256])
    if t % 4000 == 3999:
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        print('step %6d: %.4f ms per step, period16 %d, guard slowdowns %d' % (t + 1, (t1 - t0) / 4000 * 1e3, env.tuner_state()['period16'], env.tuner_state()['guard_slowdowns']), flush=True)
        t0 = t1
