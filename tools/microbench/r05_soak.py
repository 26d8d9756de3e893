"""GPU box: the tuned constants of the sweep's clock in a long run that is not bench.py's -- T steps of 65 536 full-frame envs with the episode phases
spread out, a SECOND engine (16 384 envs, full frames) taking a step on the same card every 50th step, optionally a reader of every observation byte
between two steps.  Prints the wall time of every 1 000 steps, the guard's moves, and the sweep's own time over the last 1 000 steps (library events).
    python tools/microbench/r05_soak.py [T] [reader: none|reduce32]
Race of the launch's HEAD (CW_HEAD_JOBS = 64 jobs a notch slower, cw_kernels.hip) against no head at all -- report only, there is no knob:
    make -C gym_craftingworld_amd/csrc exp EXP=-DCW_HEAD_JOBS=0 NAME=head0;  CW_LIB_PATH=$PWD/gym_craftingworld_amd/libcw_exp_head0.so python tools/microbench/r05_soak.py"""
import sys
import time
import numpy as np
import torch
sys.path.insert(0, '.')
from gym_craftingworld_amd import CraftingWorldVecEnv
T = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
reader = sys.argv[2] if len(sys.argv) > 2 else 'none'
N = 65536
env = CraftingWorldVecEnv(N, obs_mode='pixels', size=(21, 21), max_steps=300, seed=2024)
other = CraftingWorldVecEnv(16384, obs_mode='pixels', size=(21, 21), max_steps=300, seed=7)
env.reset(); other.reset()
env.set_state(step_num=((np.arange(N) * 7) % 300).astype(np.int32))
g = torch.Generator(device='cuda').manual_seed(5)
acts = torch.randint(0, 6, (512, N), device='cuda', dtype=torch.uint8, generator=g)
t_first = env.tuner_state()
print('cw_create chose', t_first, flush=True)
a = acts[0]
torch.cuda.synchronize(); t0 = time.perf_counter()
for t in range(T):
    if t == T - 1000:
        torch.cuda.synchronize()
        env.profile_begin(1000)
    env.step_async(a if reader != 'none' else acts[t % 512])
    if reader == 'reduce32':
        a = torch.remainder(env._obs.view(N, -1).view(torch.int32).sum(1, dtype=torch.int32), 6).to(torch.uint8)
    if t % 50 == 49:
        other.step_async(acts[t % 512][:16384])
    if t % 1000 == 999:
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        print('step %5d: %.4f ms per step, tuner %s' % (t + 1, (t1 - t0) / 1000 * 1e3, env.tuner_state()), flush=True)
        t0 = t1
p = env.profile_end()
frac = N * (441 + 21168) / (p['ms_render_kernel'] * 1e-3) / 8e12
print('last 1000 steps: sweep %.4f ms (median %.4f) = %.3f of the 8 TB/s peak; guard slowdowns %d; period16 %d -> %d'
      % (p['ms_render_kernel'], p['ms_render_kernel_median'], frac, env.tuner_state()['guard_slowdowns'], t_first['period16'], env.tuner_state()['period16']))
