#!/bin/bash
# GPU box: frames between 4 and 16 KiB (grids 10x10 .. 16x16) with one workgroup per CU (the rule for everything from 4 KiB up) against two and four
cd ${GRAFT_REPO_ROOT:-.}
run() { label=$1; shift
  for s in 10 12 14 16; do
    env "$@" python bench.py --quick --steps 600 --size $s 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-8s %2dx%-2d value %.4e  ms/step %.4f  sweep %.4f ms (median %.4f)  frac %.3f (median %.3f)  period16 %d' % ('$label', $s, $s, d['value'], d['ms_per_step'], r['avg_launch_ms'], r['median_launch_ms'], r['frac'], r['frac_at_median_launch'], d['tuner']['period16']))"
  done
}
run 1wg CW_X=0
run 2wg CW_TUNE_SMALL_FRAME_BYTES=16384 CW_TUNE_SMALL_BLOCKS=2
run 4wg CW_TUNE_SMALL_FRAME_BYTES=16384 CW_TUNE_SMALL_BLOCKS=4
