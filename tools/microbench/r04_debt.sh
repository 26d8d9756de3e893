#!/bin/bash
# GPU box: the clocked sweep with a catch-up allowance of D jobs (CW_TUNE_PIECE_PACE = D << 8) -- profiles/r04_clock.txt section F
cd ${GRAFT_REPO_ROOT:-.}
run() { label=$1; shift
  python bench.py --quick --steps 600 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('debt %s %-28s %.4e env-steps/s  ms/step %.4f  kernel avg %.4f med %.4f min %.4f  frac %.3f / %.3f' % ('$D', '$label', d['value'], d['ms_per_step'], r['avg_launch_ms'], r['median_launch_ms'], r['launch_ms_min_max'][0], r['frac'], r['frac_at_median_launch']))"
}
for D in ${1:-"1 2 3"}; do
  export CW_TUNE_PIECE_PACE=$((D << 8))
  run "65536 21x21"
  run "65536 21x21 desync" --desync
  run "131072 mixed desync" --envs-per-gpu 131072 --mixed-menus --desync
  run "65536 32x32 desync" --size 32 --desync
done
