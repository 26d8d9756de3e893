#!/bin/bash
# GPU box: throw-away builds UNCLOCKED (CW_TUNE_PERIOD_NS=0) on a few shapes -- r04_free.sh "<lib names>"
cd ${GRAFT_REPO_ROOT:-.}
export CW_TUNE_PERIOD_NS=0
run() { label=$1; shift
  python bench.py --quick --steps 600 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; t=d['tuner']
print('%-8s %-26s %.4e env-steps/s  ms/step %.4f  kernel avg %.4f med %.4f min %.4f max %.4f  frac %.3f / %.3f' % ('$V', '$label', d['value'], d['ms_per_step'], r['avg_launch_ms'], r['median_launch_ms'], r['launch_ms_min_max'][0], r['launch_ms_min_max'][1], r['frac'], r['frac_at_median_launch']))"
}
for V in $1; do
  [ $V = product ] && unset CW_LIB_PATH || export CW_LIB_PATH=$PWD/gym_craftingworld_amd/libcw_exp_$V.so
  run "65536 21x21"
  run "65536 21x21 desync" --desync
  run "131072 mixed" --envs-per-gpu 131072 --mixed-menus
  run "131072 mixed desync" --envs-per-gpu 131072 --mixed-menus --desync
  run "65536 32x32" --size 32
  run "65536 32x32 desync" --size 32 --desync
  run "65536 12x12" --size 12
done
