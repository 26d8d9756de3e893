#!/bin/bash
# GPU box: bench.py --quick at forced paces of the sweep, episode phases in step and spread out (profiles/r04_lookahead.txt).
#   r04_matrix.sh "0 1 2 4" "0" [extra bench args]   -> one line per (base pace in eighths, extra quarters on steps with finished envs, mode)
cd ${GRAFT_REPO_ROOT:-.}
PACES=${1:-"0 1 2 4"}; BESIDES=${2:-"0"}; shift 2
for pace in $PACES; do for beside in $BESIDES; do for mode in sync desync; do
  extra=""; [ $mode = desync ] && extra="--desync"
  CW_TUNE_PIECE_PACE=$pace CW_TUNE_PACE_BESIDE=$beside python bench.py --quick --steps 600 $extra "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('pace %s+%s %-6s  %.4e env-steps/s  ms/step %.4f  kernel avg %.4f med %.4f  frac %.3f/%.3f' % ('$pace', '$beside', '$mode', d['value'], d['ms_per_step'], r['avg_launch_ms'], r['median_launch_ms'], r['frac'], r['frac_at_median_launch']))"
done; done; done
