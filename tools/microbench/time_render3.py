"""GPU box only: what slows cw_render inside a step sequence? (buffer vs interleaving)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gym_craftingworld_amd import CraftingWorldVecEnv

N = 65536
env = CraftingWorldVecEnv(N, obs_mode='pixels_dirty', seed=0)
env.reset()
buf = torch.empty((N, 84, 84, 3), dtype=torch.uint8, device='cuda')
acts = torch.randint(0, 6, (64, N), device='cuda', dtype=torch.uint8)

def t(out, interleave, sleep_us=0):
    ts = []
    for i in range(12):
        if interleave:
            env.step_async(acts[i % 64])
        if sleep_us:
            torch.cuda._sleep(int(sleep_us * 2000))
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); env.render(out); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b))
    ts = sorted(ts[2:])
    return ts[len(ts) // 2]

print('torch buf, back-to-back      %.3f ms' % t(buf, False))
print('engine obs, back-to-back     %.3f ms' % t(env._obs, False))
print('engine init_obs, back-to-back %.3f ms' % t(env._init_img, False))
print('torch buf, after step kernels %.3f ms' % t(buf, True))
print('engine obs, after step kernels %.3f ms' % t(env._obs, True))
print('torch buf, after 300us idle   %.3f ms' % t(buf, False, 300))
