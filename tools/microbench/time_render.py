"""GPU box only: time cw_render (mode 2, caller buffer) back to back, to separate the render
kernel's own rate from in-step conditions."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gym_craftingworld_amd import CraftingWorldVecEnv

N = 65536
env = CraftingWorldVecEnv(N, obs_mode=sys.argv[1] if len(sys.argv) > 1 else 'state', seed=0)
env.reset()
out = torch.empty((N, 84, 84, 3), dtype=torch.uint8, device='cuda')
for _ in range(3):
    env.render(out)
ts = []
for _ in range(15):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); env.render(out); b.record(); b.synchronize()
    ts.append(a.elapsed_time(b))
ts.sort()
print('cw_render(ext) median %.3f ms  %.0f GB/s   min %.3f' % (ts[7], N * 21168 / ts[7] / 1e6, ts[0]))
# now steps: event around whole step
acts = torch.randint(0, 6, (64, N), device='cuda', dtype=torch.uint8)
if env.obs_mode == 'pixels':
    for t in range(10):
        env.step_async(acts[t])
    ts = []
    for t in range(30):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); env.step_async(acts[t % 64]); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    print('cw_step(pixels) median %.3f ms min %.3f' % (ts[15], ts[0]))
