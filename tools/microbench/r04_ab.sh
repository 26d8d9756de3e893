#!/bin/bash
# GPU box: A/B of throw-away builds (make -C gym_craftingworld_amd/csrc exp EXP=-D... NAME=x) against the product on a set of shapes, whatever
# CW_TUNE_* the caller exports (e.g. CW_TUNE_PERIOD_NS=0: every sweep unclocked) -- r04_ab.sh "<lib names, 'product' = the built library>"
cd ${GRAFT_REPO_ROOT:-.}
run() { label=$1; shift
  python bench.py --quick --steps 600 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; t=d['tuner']
print('%-8s %-26s %.4e env-steps/s  ms/step %.4f  kernel avg %.4f med %.4f min %.4f  frac %.3f / %.3f  period %.0f ns  beside the sweep %.1f us' % ('$V', '$label', d['value'], d['ms_per_step'], r['avg_launch_ms'], r['median_launch_ms'], r['launch_ms_min_max'][0], r['frac'], r['frac_at_median_launch'], t['period16'] / 1.6, (d['ms_per_step'] - r['avg_launch_ms'] * max(1, round(d['ms_per_step'] / r['avg_launch_ms'] - 0.3))) * 1e3))"
}
for V in $1; do
  [ $V = product ] && unset CW_LIB_PATH || export CW_LIB_PATH=$PWD/gym_craftingworld_amd/libcw_exp_$V.so
  run "65536 21x21"
  run "65536 21x21 desync" --desync
  run "131072 mixed desync" --envs-per-gpu 131072 --mixed-menus --desync
  run "65536 32x32" --size 32
  run "65536 alt 21x21" --raster alt
  run "65536 alt 21x21 desync" --raster alt --desync
  run "65536 12x12" --size 12
  run "65536 8x8" --size 8 --max-steps 100
  run "262144 8x8" --size 8 --max-steps 100 --envs-per-gpu 262144
  run "65536 5x5" --size 5 --max-steps 100
  run "262144 5x5" --size 5 --max-steps 100 --envs-per-gpu 262144
done
