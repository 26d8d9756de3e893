"""GPU box only: per-step durations of the full-frame step (one event per step, no host sync inside), as a series -- is the spread
between launches periodic, drifting, or random?   python step_series.py [n_steps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from gym_craftingworld_amd import CraftingWorldVecEnv  # noqa: E402

N = 65536
T = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
env = CraftingWorldVecEnv(N, obs_mode='pixels', seed=0, max_steps=60000)
env.reset()
acts = torch.randint(0, 6, (256, N), device='cuda', dtype=torch.uint8)
for i in range(100):
    env.step_async(acts[i % 256])
evs = [torch.cuda.Event(enable_timing=True) for _ in range(T + 1)]
evs[0].record()
for i in range(T):
    env.step_async(acts[i % 256])
    evs[i + 1].record()
torch.cuda.synchronize()
ms = np.array([evs[i].elapsed_time(evs[i + 1]) for i in range(T)])
print('steps %d  mean %.4f  median %.4f  p10 %.4f  p90 %.4f  min %.4f  max %.4f' % (T, ms.mean(), np.median(ms), np.percentile(ms, 10), np.percentile(ms, 90), ms.min(), ms.max()))
x = ms - ms.mean()
for lag in (1, 2, 3, 4, 8, 16, 32, 64):
    print('autocorrelation lag %2d: %+.3f' % (lag, float((x[:-lag] * x[lag:]).mean() / x.var())))
print('first 96 steps (us):')
for r in range(0, 96, 16):
    print(' '.join('%4.0f' % (v * 1e3) for v in ms[r:r + 16]))
print('means of consecutive blocks of 100 steps (us):', ' '.join('%.1f' % (ms[b:b + 100].mean() * 1e3) for b in range(0, T, 100)))
h, edges = np.histogram(ms * 1e3, bins=16)
print('histogram (us):', ' '.join('%.0f:%d' % (edges[i], h[i]) for i in range(len(h))))
