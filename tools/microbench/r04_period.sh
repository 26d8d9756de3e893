#!/bin/bash
# GPU box: the clocked sweep -- bench.py --quick over job periods (ns), phases in step and spread out, product or throw-away builds side by side
#   r04_period.sh "<lib names, 'product' = the built library>" "<periods ns>" "<values of CW_EXP_KNOB, an experiment build's own knob; 0>" "<modes>" [extra bench args]
cd ${GRAFT_REPO_ROOT:-.}
for v in $1; do
  [ $v = product ] && unset CW_LIB_PATH || export CW_LIB_PATH=$PWD/gym_craftingworld_amd/libcw_exp_$v.so
  for per in $2; do for knob in $3; do for mode in $4; do
    extra=""; [ $mode = desync ] && extra="--desync"
    CW_TUNE_PERIOD_NS=$per CW_EXP_KNOB=$knob python bench.py --quick --steps 600 $extra $5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$v period %4s ns knob %s %-6s  %.4e env-steps/s  ms/step %.4f  kernel avg %.4f med %.4f min %.4f  frac %.3f/%.3f' % ('$per', '$knob', '$mode', d['value'], d['ms_per_step'], r['avg_launch_ms'], r['median_launch_ms'], r['launch_ms_min_max'][0], r['frac'], r['frac_at_median_launch']))"
  done; done; done
done
