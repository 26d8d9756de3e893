"""GPU box, under rocprofv3 --kernel-trace --stats: 300 full-frame steps of 65 536 envs with and without auto-reset -- how long does the step kernel
take right after a sweep (cw_step_fused_kernel vs the lane-per-env cw_step_kernel)?"""
import sys
import torch
sys.path.insert(0, '.')
from gym_craftingworld_amd import CraftingWorldVecEnv
N = 65536
acts = torch.randint(0, 6, (64, N), device='cuda', dtype=torch.uint8)
for auto in (True, False):
    e = CraftingWorldVecEnv(N, obs_mode='pixels', auto_reset=auto, seed=1)
    e.reset()
    for t in range(290):
        e.step_async(acts[t % 64])
    torch.cuda.synchronize()
    e.close()
