#!/bin/bash
# GPU box: throw-away builds on the shapes whose steady state (episode phases spread out) is the fragile one -- r04_desync.sh "<lib names>"
cd ${GRAFT_REPO_ROOT:-.}
run() { label=$1; shift
  python bench.py --quick --steps 600 --desync "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; t=d['tuner']
print('%-8s %-26s %.4e env-steps/s  ms/step %.4f  kernel avg %.4f med %.4f min %.4f max %.4f  frac %.3f / %.3f  period %.0f / %.0f ns' % ('$V', '$label', d['value'], d['ms_per_step'], r['avg_launch_ms'], r['median_launch_ms'], r['launch_ms_min_max'][0], r['launch_ms_min_max'][1], r['frac'], r['frac_at_median_launch'], t['period16'] / 1.6, t['period16_busy'] / 1.6))"
}
for V in $1; do
  [ $V = product ] && unset CW_LIB_PATH || export CW_LIB_PATH=$PWD/gym_craftingworld_amd/libcw_exp_$V.so
  run "65536 21x21 desync"
  run "131072 mixed desync" --envs-per-gpu 131072 --mixed-menus
  run "65536 alt desync" --raster alt
  run "65536 32x32 desync" --size 32
done
