"""GPU box: is the guard RIGHT to give way when another engine's steps run between this engine's sweeps?  The loop of soak_long.py (65 536 full-frame envs
and the dirty-cell engine of the same batch stepped alternately on one stream, phases spread out), three times: guard on (default), guard off at cw_create's
rate, guard off at a forced 6.5 TB/s.  After WARM steps: the sweep's time by the library's events and the wall time per iteration over 2 000 steps.
python tools/microbench/r05_guard_two_engines.py [WARM]"""
import os
import sys
import time
import numpy as np
import torch
sys.path.insert(0, '.')
from gym_craftingworld_amd import CraftingWorldVecEnv
WARM = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
N = 65536
kw = dict(size=(21, 21), max_steps=300, seed=2024)
g = torch.Generator(device='cuda').manual_seed(5)
acts = torch.randint(0, 6, (512, N), device='cuda', dtype=torch.uint8, generator=g)
phase = ((np.arange(N) * 7) % 300).astype(np.int32)
for name, env in (('guard on', {}), ('guard off', {'CW_TUNE_GUARD': '0'}), ('guard off, 6.5 TB/s', {'CW_TUNE_GUARD': '0', 'CW_TUNE_RATE_TBS': '6.5'}),
                  ('guard on, alone', {})):
    for k in ('CW_TUNE_GUARD', 'CW_TUNE_RATE_TBS'):
        os.environ.pop(k, None)
    os.environ.update(env)
    full = CraftingWorldVecEnv(N, obs_mode='pixels', **kw)
    for k in env:
        os.environ.pop(k, None)
    alone = name.endswith('alone')
    dirty = None if alone else CraftingWorldVecEnv(N, obs_mode='pixels_dirty', **kw)
    full.reset(); full.set_state(step_num=phase)
    if dirty is not None:
        dirty.reset(); dirty.set_state(step_num=phase)
    t0s = full.tuner_state()
    for t in range(WARM):
        a = acts[t % 512]
        full.step_async(a)
        if dirty is not None:
            dirty.step_async(a)
        if t % 4096 == 4095:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    t1s = full.tuner_state()
    full.profile_begin(2000)
    t0 = time.perf_counter()
    for t in range(2000):
        a = acts[t % 512]
        full.step_async(a)
        if dirty is not None:
            dirty.step_async(a)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 2000 * 1e3
    p = full.profile_end()
    print('%-22s period16 %4d -> %4d (slowdowns %2d)  sweep %.4f ms (median %.4f, max %.4f)  iteration %.4f ms'
          % (name, t0s['period16'], t1s['period16'], t1s['guard_slowdowns'], p['ms_render_kernel'], p['ms_render_kernel_median'], p['ms_render_kernel_max'], wall), flush=True)
    full.close()
    if dirty is not None:
        dirty.close()
