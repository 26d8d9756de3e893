import os, sys, time
sys.path.insert(0, '.')
import torch
from gym_craftingworld_amd import CraftingWorldVecEnv
N, T, S = 65536, 2000, 8
acts = torch.randint(0, 4, (256, N), device='cuda', dtype=torch.uint8)
env = CraftingWorldVecEnv(N, obs_mode='state', size=(S, S), max_steps=300, seed=1, reward_style='subset', selected_tasks=['EatBread'], number_of_tasks=1)
env.reset()
for t in range(T):
    env.step_async(acts[t % 256])
torch.cuda.synchronize()
env.close()
