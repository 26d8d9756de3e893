#!/bin/bash
# GPU box: workgroups per CU for the sweep of small frames (CW_TUNE_SMALL_BLOCKS), both painters
cd ${GRAFT_REPO_ROOT:-.}
run() { label=$1; raster=$2; sizes=$3; shift 3
  for s in $sizes; do
    env "$@" python bench.py --quick --steps 600 --size $s --raster $raster 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-12s %-3s %2dx%-2d value %.4e  ms/step %.4f  sweep %.4f ms (median %.4f)  frac %.3f (median %.3f)  period16 %d' % ('$label', '$raster', $s, $s, d['value'], d['ms_per_step'], r['avg_launch_ms'], r['median_launch_ms'], r['frac'], r['frac_at_median_launch'], d['tuner']['period16']))"
  done
}
for wg in 1 2 4 8; do
  run pieces_${wg}wg ray "4 5 6 7 8 9" CW_TUNE_GATHER=0 CW_TUNE_SMALL_BLOCKS=$wg
  run gather_${wg}wg ray "4 5 6 7" CW_TUNE_SMALL_BLOCKS=$wg
  run pieces_${wg}wg alt "5 8" CW_TUNE_SMALL_BLOCKS=$wg
done
