#!/bin/bash
# GPU box: cw_create's pick of the sweep's clock in FRESH processes (a bench run's situation), with the table it decided on (CW_TUNE_VERBOSE: median / 90th
# percentile / mean of 20 launches per candidate), and what the 600 sweeps after it take; then the same with the rate forced to 7.7 and 7.4 TB/s.
#   bash tools/microbench/r06_create_pick.sh [processes]
cd ${GRAFT_REPO_ROOT:-.}
R=${1:-8}
run() {
  CW_TUNE_VERBOSE=1 python - <<PY 2>&1 | grep -v amdgpu.ids
import sys
sys.path.insert(0, '.')
import torch
from gym_craftingworld_amd import CraftingWorldVecEnv
N = 65536
env = CraftingWorldVecEnv(N, obs_mode='pixels', size=(21, 21), max_steps=300, seed=1)
acts = torch.randint(0, 6, (64, N), device='cuda', dtype=torch.uint8)
env.reset()
for t in range(120):
    env.step_async(acts[t % 64])
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
env.profile_begin(600)
for t in range(600):
    env.step_async(acts[t % 64])
torch.cuda.synchronize()
dt = time.perf_counter() - t0
p = env.profile_end()
print('$1 period16 %d: 600 steps %.4f ms per step, sweeps %.4f ms (median %.4f), guard %d' % (env.tuner_state()['period16'], dt / 600 * 1e3, p['ms_render_kernel'], p['ms_render_kernel_median'], env.tuner_state()['guard_slowdowns']))
env.close()
PY
}
for r in $(seq 1 $R); do
  unset CW_TUNE_PERIOD_NS; run "picked "
  CW_TUNE_PERIOD_NS=545 run "forced 7.7"
  CW_TUNE_PERIOD_NS=567 run "forced 7.4"
done
