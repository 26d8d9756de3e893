#!/bin/bash
# GPU box: the head's two notches (a launch's first 64 jobs run CW_HEAD_NOTCH TB/s slower, CW_BUSY_NOTCH after a step on which envs finished) now that the
# step kernel paints the finished envs' frames with four waves: can the busy head be shorter?   forced clock 545 ns, phases spread out and in step, two repetitions
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2; do for n in "0.4 0.75" "0.4 0.55" "0.4 0.4" "0.25 0.55" "0.25 0.25" "0.0 0.0"; do set -- $n
  for mode in desync sync; do
    extra=""; [ $mode = desync ] && extra="--desync"
    CW_TUNE_PERIOD_NS=545 CW_TUNE_HEAD_NOTCH=$1 CW_TUNE_BUSY_NOTCH=$2 python bench.py --quick --steps 600 $extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('head %s busy %s %-6s value %.4e  ms/step %.4f  sweep %.4f ms (median %.4f, max %.4f) frac %.3f' % ('$1', '$2', '$mode', d['value'], d['ms_per_step'], r['avg_launch_ms'], r['median_launch_ms'], r['launch_ms_min_max'][1], r['frac']))"
  done
done; done
