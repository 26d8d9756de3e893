"""GPU box only: cw_render queued continuously takes 0.2235-0.2258 ms, inside cw_step 0.235-0.243.  Which neighbour costs the difference?
Every variant queues 300 x [neighbour; event; cw_render; event] with no host sync inside and reports the median render time.
   CW_TUNE_RENDER_PACE=257 python instep_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from gym_craftingworld_amd import CraftingWorldVecEnv  # noqa: E402

N = 65536
R = CraftingWorldVecEnv(N, obs_mode='state', seed=0)
R.reset()
S = CraftingWorldVecEnv(N, obs_mode='state', seed=1, max_steps=60000)
S.reset()
D = CraftingWorldVecEnv(N, obs_mode='pixels_dirty', seed=2, max_steps=60000)
D.reset()
out = torch.empty((N, 84, 84, 3), dtype=torch.uint8, device='cuda')
acts = torch.randint(0, 6, (64, N), device='cuda', dtype=torch.uint8)
small = torch.zeros(1 << 20, device='cuda')
big = torch.zeros(64 << 20, device='cuda', dtype=torch.uint8)
side = torch.cuda.Stream()


def probe(label, pre, n=300):
    evs = []
    for i in range(n):
        if pre:
            pre(i)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); R.render(out); b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in evs[50:])
    print('%-64s render median %.4f ms  (p10 %.4f  p90 %.4f)' % (label, ms[len(ms) // 2], ms[len(ms) // 10], ms[len(ms) * 9 // 10]))


def forkjoin(i):
    e = torch.cuda.Event()
    e.record()
    side.wait_event(e)
    with torch.cuda.stream(side):
        small.add_(1)
    e2 = torch.cuda.Event()
    e2.record(side)
    torch.cuda.current_stream().wait_event(e2)


for rep in range(2):
    probe('nothing between launches', None)
    probe('a 4-MB torch kernel between launches', lambda i: small.add_(1))
    probe('a 64-MB torch kernel (reads + writes 128 MB) between launches', lambda i: big.add_(1))
    probe('a state-only engine step (fused kernel, 65536 envs) between launches', lambda i: S.step_async(acts[i % 64]))
    probe('a dirty-cell engine step between launches', lambda i: D.step_async(acts[i % 64]))
    probe('fork: a 4-MB kernel on a second stream, joined before the render', forkjoin)
