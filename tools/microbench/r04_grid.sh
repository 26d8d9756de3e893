#!/bin/bash
# GPU box: the clocked sweep with 1/2, 1, 2, 3 workgroups per CU (make exp EXP=-DCW_EXP_GRID=1 NAME=grid; the period follows the number of waves)
cd ${GRAFT_REPO_ROOT:-.}
export CW_LIB_PATH=$PWD/gym_craftingworld_amd/libcw_exp_grid.so
for h in 2 1 4 6; do for rate in 7.0 7.2; do for mode in sync desync; do
  extra=""; [ $mode = desync ] && extra="--desync"
  CW_EXP_GRID_HALVES=$h CW_TUNE_RATE_TBS=$rate python bench.py --quick --steps 600 $extra $1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; t=d['tuner']
print('wg/cu %.1f rate %s %-6s  %.4e env-steps/s  ms/step %.4f  kernel avg %.4f med %.4f min %.4f  frac %.3f/%.3f  period %.0f ns' % ($h / 2.0, '$rate', '$mode', d['value'], d['ms_per_step'], r['avg_launch_ms'], r['median_launch_ms'], r['launch_ms_min_max'][0], r['frac'], r['frac_at_median_launch'], t['period16'] / 1.6))"
done; done; done
