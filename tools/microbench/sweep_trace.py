"""GPU box: where a launch of the clocked sweep spends the time beside its schedule.  Needs a trace build of the library
(git apply the patch quoted in profiles/r04_clock.txt H; make exp EXP=-DCW_EXP_SWEEP_TRACE=1 NAME=strace; CW_LIB_PATH=.../libcw_exp_strace.so):
every wave leaves its entry time, the time its last store was issued and had drained, and what its clock forgave."""
import ctypes as C
import sys
import numpy as np
import torch
from gym_craftingworld_amd import CraftingWorldVecEnv

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
size = int(sys.argv[2]) if len(sys.argv) > 2 else 21
desync = len(sys.argv) > 3 and sys.argv[3] == 'desync'
env = CraftingWorldVecEnv(n, size=(size, size), obs_mode='pixels', device='cuda:0')
env.seed(0)
env.reset()
g = torch.Generator(device='cuda').manual_seed(1)
acts = torch.randint(0, 6, (400, n), device='cuda', dtype=torch.int32, generator=g)
if desync:
    env.set_state(step_num=((np.arange(n) * 7) % 300).astype(np.int32))
for t in range(250):
    env.step_async(acts[t]); env.step_wait()
torch.cuda.synchronize()
period = env.tuner_state()['period16'] / 16.0            # ticks of 10 ns
buf = (C.c_ulonglong * (1024 * 8))()
for rep in range(12):
    for t in range(7):                                    # back to back: the LAST launch's stamps are read (an idle gap before a launch flatters it)
        env.step_async(acts[300 + (7 * rep + t) % 90]); env.step_wait()
    torch.cuda.synchronize()
    assert env._lib.cwk_trace_read(buf) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 8).astype(np.int64)
    t_in, t_issue, t_drain, n_forg, first, last = a[:, 0], a[:, 2], a[:, 3], a[:, 4], a[:, 5], a[:, 6]
    z = t_in.min()
    frame_bytes = 48 * size * size
    jobs = -(-(n * frame_bytes) // 4096)
    q = -(-jobs // 1024)
    had = first >= 0
    print('launch %2d: period %.1f ns, %d jobs/wave, schedule %.1f us | last store issued med %6.2f max %6.2f us | debts forgiven per wave med %d max %d, waves with none %d; '
          'first at job med %d (min %d), last at job med %d' %
          (rep, period * 10, q, q * period / 100, np.median(t_issue - z) / 100, (t_issue - z).max() / 100, np.median(n_forg), n_forg.max(), (~had).sum(),
           np.median(first[had]) if had.any() else -1, first[had].min() if had.any() else -1, np.median(last[had]) if had.any() else -1))
