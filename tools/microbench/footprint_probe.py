"""GPU box only: does cw_render's launch time depend on how much OTHER memory the process has written?  (time_render.py placement: the
first ~30 launches of a process run at 0.225 ms, later ones at 0.28; clock_trace.py with one buffer: 0.2235 for seconds.)
   CW_TUNE_RENDER_PACE=257 python footprint_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from gym_craftingworld_amd import CraftingWorldVecEnv  # noqa: E402

N = 65536
R = CraftingWorldVecEnv(N, obs_mode='state', seed=0)
R.reset()
out = torch.empty((N, 84, 84, 3), dtype=torch.uint8, device='cuda')


def probe(label, buf=None, n=300):
    buf = out if buf is None else buf
    evs = []
    for i in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); R.render(buf); b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in evs[50:])
    print('%-72s render median %.4f ms  (p10 %.4f  p90 %.4f)' % (label, ms[len(ms) // 2], ms[len(ms) // 10], ms[len(ms) * 9 // 10]), flush=True)


probe('fresh process, one 1.4-GB buffer')
probe('again')
if len(sys.argv) > 1 and sys.argv[1] == 'short':
    raise SystemExit(0)
others = []
for gb in (1.4, 2.8, 4.2, 8.4, 16.8):
    while sum(o.numel() for o in others) < gb * 1e9:
        others.append(torch.zeros((N, 84, 84, 3), dtype=torch.uint8, device='cuda'))
    torch.cuda.synchronize()
    probe('after allocating + zeroing %.1f GB of other buffers' % (sum(o.numel() for o in others) / 1e9))
probe('rendering into the LAST allocated buffer', others[-1])
probe('rendering into the first buffer again')
del others
torch.cuda.empty_cache()
probe('other buffers freed (empty_cache)')
x = torch.zeros((N, 84, 84, 3), dtype=torch.uint8, device='cuda')
for i in range(200):
    x.add_(1)
torch.cuda.synchronize()
probe('after 200 read-modify-write passes over another 1.4-GB buffer')
import time
time.sleep(2.0)
probe('after 2 s of idleness')
