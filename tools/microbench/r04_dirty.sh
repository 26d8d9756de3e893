#!/bin/bash
# GPU box: the dirty-cell mode on its own -- bench.py --obs-mode pixels_dirty, episode phases in step and spread out
cd ${GRAFT_REPO_ROOT:-.}
for mode in sync desync; do
  extra=""; [ $mode = desync ] && extra="--desync"
  for rep in 1 2; do
  python bench.py --quick --steps 2000 --obs-mode pixels_dirty $extra $1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('dirty $mode  %.4e env-steps/s  us/step %.2f' % (d['value'], d['ms_per_step'] * 1e3))"
  done
done
