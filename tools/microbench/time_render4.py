"""GPU box only: which preceding kernel slows cw_render?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gym_craftingworld_amd import CraftingWorldVecEnv
N = 65536
buf = torch.empty((N, 84, 84, 3), dtype=torch.uint8, device='cuda')
acts = torch.randint(0, 6, (64, N), device='cuda', dtype=torch.uint8)
small = torch.zeros(1 << 20, device='cuda')

def t(env, pre):
    ts = []
    for i in range(12):
        pre(i)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); env.render(buf); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b))
    ts = sorted(ts[2:])
    return ts[len(ts) // 2]

for mode, ar in (('state', False), ('state', True), ('pixels_dirty', True)):
    env = CraftingWorldVecEnv(N, obs_mode=mode, seed=0, auto_reset=ar)
    env.reset()
    print('%-13s auto_reset=%d: none %.3f | torch add kernel %.3f | step %.3f' % (
        mode, ar, t(env, lambda i: None), t(env, lambda i: small.add_(1)), t(env, lambda i: env.step_async(acts[i % 64]))))
    env.close()
