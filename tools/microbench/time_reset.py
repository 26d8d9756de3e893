"""GPU box: an explicit reset() of the whole batch (full-frame mode): the reset kernel (every env takes its look-ahead record, or computes its
episode if it has none) + the sweeps of the three frame arrays; the refill of the next records follows on the stream.  python tools/microbench/time_reset.py [envs]"""
import sys
import torch
sys.path.insert(0, '.')
from gym_craftingworld_amd import CraftingWorldVecEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
env = CraftingWorldVecEnv(n, size=(21, 21), obs_mode='pixels', device='cuda:0')
env.seed(0)
a, b, c = (torch.cuda.Event(enable_timing=True) for _ in range(3))
for rep in range(4):
    torch.cuda.synchronize()
    a.record(); env.reset(); b.record()
    acts = torch.zeros(n, dtype=torch.uint8, device='cuda')
    env.step_async(acts); env.step_wait(); c.record()
    torch.cuda.synchronize()
    print('reset %d: %.3f ms until the frames are there and the next records computed; the first step after it %.3f ms' % (rep, a.elapsed_time(b), b.elapsed_time(c)))
