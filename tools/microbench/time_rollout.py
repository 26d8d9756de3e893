"""GPU box only: throughput of cw_rollout (persistent multi-step kernel) vs per-step launches, state-only mode."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gym_craftingworld_amd import CraftingWorldVecEnv

for N in (4096, 65536, 1048576):
    T = 600
    acts = torch.randint(0, 6, (T, N), device='cuda', dtype=torch.uint8)
    for record in (False, True):
        env = CraftingWorldVecEnv(N, obs_mode='state', seed=0)
        env.reset()
        env.rollout(acts[:50], record=record)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        env.rollout(acts, record=record)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print('N=%8d rollout T=%d record=%d: %.3e env-steps/s (%.2f us/step)' % (N, T, record, N * T / dt, dt / T * 1e6))
        env.close()
    env = CraftingWorldVecEnv(N, obs_mode='state', seed=0)
    env.reset()
    for t in range(50):
        env.step_async(acts[t])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(T):
        env.step_async(acts[t])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print('N=%8d per-step launches     : %.3e env-steps/s (%.2f us/step)' % (N, N * T / dt, dt / T * 1e6))
    env.close()
