"""GPU box: what does the look-ahead refill period cost when episodes are much SHORTER than max_steps -- a policy that succeeds?  Stand-in for a competent
policy: 8x8 grids, reward_style='subset', the one task EatBread (walk onto the bread), random actions: episodes of a few dozen steps under max_steps = 300,
whose static refill period is 64 steps.  'default' = the product (round 6: the period adapts to the slow-path resets it sees); per forced period (CW_TUNE_LA_PERIOD): us per step, episodes finished per step, resets taken the slow way per step.
    python tools/microbench/r06_short_episodes.py [obs_mode] [size]"""
import os, sys, time
sys.path.insert(0, '.')
import torch
from gym_craftingworld_amd import CraftingWorldVecEnv
mode = sys.argv[1] if len(sys.argv) > 1 else 'state'
S = int(sys.argv[2]) if len(sys.argv) > 2 else 8
N, T = 65536, 3000
acts = torch.randint(0, 4, (256, N), device='cuda', dtype=torch.uint8)        # moves only: a walker
for period in ('default', '64', '32', '16', '8', '4', 'default'):
    if period == 'default':
        os.environ.pop('CW_TUNE_LA_PERIOD', None)
    else:
        os.environ['CW_TUNE_LA_PERIOD'] = period
    env = CraftingWorldVecEnv(N, obs_mode=mode, size=(S, S), max_steps=300, seed=1, reward_style='subset', selected_tasks=['EatBread'], number_of_tasks=1)
    env.reset()
    for t in range(600):
        env.step_async(acts[t % 256])
    torch.cuda.synchronize()
    c0 = env._counters_raw.cpu().clone()
    t0 = time.perf_counter()
    for t in range(T):
        env.step_async(acts[t % 256])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    c1 = env._counters_raw.cpu()
    print('%-12s S=%d period %-7s: %.2f us/step, %.3e env-steps/s; episodes finished per step %.0f (mean length %.1f), slow-path resets per step %.1f' % (
        mode, S, period, dt / T * 1e6, N * T / dt, float(c1[1] - c0[1]) / T, N * T / max(float(c1[1] - c0[1]), 1), float(c1[5] - c0[5]) / T), flush=True)
    env.close()
