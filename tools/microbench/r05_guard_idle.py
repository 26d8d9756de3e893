"""GPU box: what makes the guard lower the clock in bench.py's soak (beside the CPU baseline, the host waiting for the card every 4 096 steps)?
Two loops of 49 152 steps of the headline batch, phases spread out, CW_TUNE_VERBOSE=1: (a) never waiting, (b) waiting every 4 096 steps.
(A third loop with 15 pure-Python threads spinning beside it starved this thread of the GIL and was killed for silence: do not add it back.)
python tools/microbench/r05_guard_idle.py"""
import sys
import threading
import time
import numpy as np
import torch
sys.path.insert(0, '.')
from gym_craftingworld_amd import CraftingWorldVecEnv
N, T = 65536, 49152
acts = torch.randint(0, 6, (512, N), device='cuda', dtype=torch.uint8, generator=torch.Generator(device='cuda').manual_seed(5))
phase = ((np.arange(N) * 7) % 300).astype(np.int32)
stop = False


def spin():
    x = 1.0
    while not stop:
        for _ in range(100000):
            x = x * 1.0000001 + 1e-9


for name, every, busy in (('never waiting', 0, 0), ('waiting every 4096 steps', 4096, 0)):
    env = CraftingWorldVecEnv(N, obs_mode='pixels', size=(21, 21), max_steps=300, seed=2024)
    env.reset(); env.set_state(step_num=phase)
    for t in range(600):
        env.step_async(acts[t % 512])
    torch.cuda.synchronize()
    stop = False
    ths = [threading.Thread(target=spin, daemon=True) for _ in range(busy)]
    for th in ths:
        th.start()
    t0s = env.tuner_state()
    print('==', name, flush=True)
    sys.stderr.write('== %s\n' % name); sys.stderr.flush()
    t0 = time.perf_counter()
    for t in range(T):
        env.step_async(acts[t % 512])
        if every and t % every == every - 1:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    stop = True
    for th in ths:
        th.join()
    t1s = env.tuner_state()
    env.profile_begin(300)
    for t in range(300):
        env.step_async(acts[t % 512])
    torch.cuda.synchronize()
    p = env.profile_end()
    print('%-44s %.4f ms per step; clock %d -> %d, %d slowdowns; sweep afterwards %.4f ms' % (name, dt / T * 1e3, t0s['period16'], t1s['period16'], t1s['guard_slowdowns'], p['ms_render_kernel']), flush=True)
    env.close()
