#!/bin/bash
# (needs the two-line CW_TUNE_CALIB_WARM hook described in profiles/r06_create_warm.txt: not in the product)
# GPU box: does cw_create pick a slower clock because it measures its candidates on a card that has just idled through process start-up?  Fresh processes,
# alternating: the product's calibration (one uncounted round of 24 launches, then 7.7 TB/s first) against CW_TUNE_CALIB_WARM=4 / 12 (4 / 12 more uncounted
# rounds: ~20 / ~60 ms).  Each process: create, CW_TUNE_VERBOSE table, 100 + 600 steps, the sweep's time by the library's events.
#   bash tools/microbench/r06_create_warm.sh [rounds]  -> stdout
cd ${GRAFT_REPO_ROOT:-.}
R=${1:-5}
for r in $(seq 1 $R); do
  for warm in 0 4 12; do
    if [ $warm = 0 ]; then unset CW_TUNE_CALIB_WARM; else export CW_TUNE_CALIB_WARM=$warm; fi
    CW_TUNE_VERBOSE=1 python - <<PY 2>&1 | grep -v amdgpu.ids | sed "s/^/warm=$warm /"
import sys, time
sys.path.insert(0, '.')
import torch
from gym_craftingworld_amd import CraftingWorldVecEnv
N = 65536
t0 = time.perf_counter()
env = CraftingWorldVecEnv(N, obs_mode='pixels', size=(21, 21), max_steps=300, seed=1)
t_create = time.perf_counter() - t0
acts = torch.randint(0, 6, (64, N), device='cuda', dtype=torch.uint8)
env.reset()
for t in range(100):
    env.step_async(acts[t % 64])
torch.cuda.synchronize()
env.profile_begin(600)
for t in range(600):
    env.step_async(acts[t % 64])
torch.cuda.synchronize()
p = env.profile_end()
print('picked period16 %d in %.2f s; then 600 sweeps %.4f ms (median %.4f); guard %d' % (env.tuner_state()['period16'], t_create, p['ms_render_kernel'], p['ms_render_kernel_median'], env.tuner_state()['guard_slowdowns']))
env.close()
PY
  done
done
