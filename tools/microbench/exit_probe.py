import sys, os
sys.path.insert(0, os.getcwd())
import torch
from gym_craftingworld_amd import CraftingWorldVecEnv
env = CraftingWorldVecEnv(8, size=(5, 5), obs_mode='pixels')
obs = env.reset()
keep = obs['observation']          # a view that outlives everything
raise SystemExit(3)                # exit without close(), views alive
