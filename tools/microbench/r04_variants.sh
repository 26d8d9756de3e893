#!/bin/bash
# GPU box: throw-away builds gym_craftingworld_amd/libcw_exp_<name>.so side by side -- cw_create's pace table (20 launches per pace) and two
# bench.py --quick runs each (phases in step / spread out) at forced paces
#   r04_variants.sh "a b c" "2 8"
cd ${GRAFT_REPO_ROOT:-.}
for v in $1; do
  export CW_LIB_PATH=$PWD/gym_craftingworld_amd/libcw_exp_$v.so
  echo "== variant $v"
  CW_TUNE_VERBOSE=1 python -c "
import torch
from gym_craftingworld_amd import CraftingWorldVecEnv
e = CraftingWorldVecEnv(65536, obs_mode='pixels', seed=0); e.close()" 2>&1 | grep "sweep pace"
  for pace in $2; do for mode in sync desync; do
    extra=""; [ $mode = desync ] && extra="--desync"
    CW_TUNE_PIECE_PACE=$pace CW_TUNE_PACE_BESIDE=0 python bench.py --quick --steps 600 $extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$v pace %s %-6s  %.4e env-steps/s  ms/step %.4f  kernel avg %.4f med %.4f  frac %.3f/%.3f' % ('$pace', '$mode', d['value'], d['ms_per_step'], r['avg_launch_ms'], r['median_launch_ms'], r['frac'], r['frac_at_median_launch']))"
  done; done
done
