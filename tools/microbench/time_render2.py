"""GPU box only: does cw_render's rate depend on which buffer it writes (placement) within one process?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gym_craftingworld_amd import CraftingWorldVecEnv

N = 65536
env = CraftingWorldVecEnv(N, obs_mode='state', seed=0)
env.reset()
bufs = [torch.empty((N, 84, 84, 3), dtype=torch.uint8, device='cuda') for _ in range(6)]
def t(out):
    for _ in range(2):
        env.render(out)
    ts = []
    for _ in range(9):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); env.render(out); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[4]
for rep in range(2):
    for i, o in enumerate(bufs):
        print('buf %d addr %#x  %.3f ms' % (i, o.data_ptr(), t(o)))
sys.exit(0)
big = torch.empty((N + 64) * 21168, dtype=torch.uint8, device='cuda')
for sh in (0, 1, 2, 3, 5, 8, 16, 32):
    o = big[sh * 21168: (sh + N) * 21168].view(N, 84, 84, 3)
    print('shift %2d frames addr %#x  %.3f ms' % (sh, o.data_ptr(), t(o)))
