#!/bin/bash
# GPU box: shapes whose sweep is several launches per step (32x32: 2, 64x64: 2, 2^20 envs: 8) -- r04_chunks.sh "<lib names>"
cd ${GRAFT_REPO_ROOT:-.}
run() { label=$1; shift
  python bench.py --quick --steps 600 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; t=d['tuner']
print('%-8s %-26s %.4e env-steps/s  ms/step %.4f  kernel avg %.4f med %.4f  frac %.3f / %.3f  period %.0f ns' % ('$V', '$label', d['value'], d['ms_per_step'], r['avg_launch_ms'], r['median_launch_ms'], r['frac'], r['frac_at_median_launch'], t['period16'] / 1.6))"
}
for V in $1; do
  [ $V = product ] && unset CW_LIB_PATH || export CW_LIB_PATH=$PWD/gym_craftingworld_amd/libcw_exp_$V.so
  run "65536 32x32" --size 32
  run "65536 32x32 desync" --size 32 --desync
  run "16384 64x64" --size 64 --envs-per-gpu 16384
  run "1048576 21x21" --envs-per-gpu 1048576
  run "1048576 21x21 desync" --envs-per-gpu 1048576 --desync
done
