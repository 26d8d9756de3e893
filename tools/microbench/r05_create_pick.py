"""GPU box: what does cw_create's calibration of the sweep's clock see, create after create?  CW_TUNE_VERBOSE logs median / 90th percentile / mean of
20 launches per candidate rate; ten engines in a row (each one followed by 300 steps so that the next calibration starts from a working card).
    CW_TUNE_VERBOSE=1 python tools/microbench/r05_create_pick.py [n] [desync]"""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from gym_craftingworld_amd import CraftingWorldVecEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
N = 65536
acts = torch.randint(0, 6, (64, N), device='cuda', dtype=torch.uint8)
for i in range(n):
    env = CraftingWorldVecEnv(N, obs_mode='pixels', size=(21, 21), max_steps=300, seed=i)
    env.reset()
    if len(sys.argv) > 2:
        env.set_state(step_num=((np.arange(N) * 7) % 300).astype(np.int32))
    for t in range(100):
        env.step_async(acts[t % 64])
    torch.cuda.synchronize()
    env.profile_begin(300)
    for t in range(300):
        env.step_async(acts[t % 64])
    torch.cuda.synchronize()
    p = env.profile_end()
    print('engine %d: period16 %d, then 300 sweeps %.4f ms (median %.4f)' % (i, env.tuner_state()['period16'], p['ms_render_kernel'], p['ms_render_kernel_median']), flush=True)
    env.close()
