#!/bin/bash
# GPU box: with a consumer between two steps the write path has time to drain -- does the sweep take a FASTER clock than back to back (7.7 TB/s = 545 ns)?
cd ${GRAFT_REPO_ROOT:-.}
for c in reduce32 reduce; do for ns in default 0 560 545 530 515 500 480; do
  if [ $ns = default ]; then unset CW_TUNE_PERIOD_NS; else export CW_TUNE_PERIOD_NS=$ns; fi
  python bench.py --quick --steps 300 --desync --consumer $c --consumer-steps 400 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=d['policy_in_loop']['$c']; s=b['sweep']
print('consumer %-8s period %-7s back to back: sweep %.4f ms frac %.3f | in the loop: sweep %.4f (median %.4f, max %.4f) frac %.3f, env part %.4f ms' % ('$c', '$ns', d['roofline']['avg_launch_ms'], d['roofline']['frac'], s['avg_launch_ms'], s['median_launch_ms'], s['launch_ms_min_max'][1], s['frac'], b['env_ms_per_step']))"
done; done
