#!/bin/bash
# GPU box: the clocked sweep -- bench.py --quick over job periods (ns) and intra-job paces, phases in step and spread out
#   r04_period.sh "<lib names>" "<periods ns>" "<paces>" "<modes>" [extra bench args]
cd ${GRAFT_REPO_ROOT:-.}
for v in $1; do
  [ $v = product ] && unset CW_LIB_PATH || export CW_LIB_PATH=$PWD/gym_craftingworld_amd/libcw_exp_$v.so
  for per in $2; do for pace in $3; do for mode in $4; do
    extra=""; [ $mode = desync ] && extra="--desync"
    CW_TUNE_PERIOD_NS=$per CW_TUNE_PIECE_PACE=$pace CW_TUNE_PACE_BESIDE=0 python bench.py --quick --steps 600 $extra $5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$v period %4s ns pace %s %-6s  %.4e env-steps/s  ms/step %.4f  kernel avg %.4f med %.4f min %.4f  frac %.3f/%.3f' % ('$per', '$pace', '$mode', d['value'], d['ms_per_step'], r['avg_launch_ms'], r['median_launch_ms'], r['launch_ms_min_max'][0], r['frac'], r['frac_at_median_launch']))"
  done; done; done
done
