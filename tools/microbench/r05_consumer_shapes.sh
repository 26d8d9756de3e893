#!/bin/bash
# GPU box: a reader of every observation byte between two steps (bench.py --consumer reduce32) on the other BASELINE shapes
cd ${GRAFT_REPO_ROOT:-.}
run() { label=$1; shift
  python bench.py --quick --steps 300 --desync --consumer reduce32 --consumer-steps 300 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=list(d['policy_in_loop'].values())[0]; s=b['sweep']
print('%-26s back to back: sweep %.4f ms frac %.3f | reader in the loop (%.3f ms, %.0f GB/s): sweep %.4f ms (median %.4f) frac %.3f, env part %.4f ms, guard moves %d' % ('$label', d['roofline']['avg_launch_ms'], d['roofline']['frac'], b['consumer_ms'], b['consumer_read_GBs'], s['avg_launch_ms'], s['median_launch_ms'], s['frac'], b['env_ms_per_step'], b['guard_moves']))"
}
run "21x21 (headline)"
run "32x32 (configs[4])" --size 32
run "131072 mixed (configs[3])" --envs-per-gpu 131072 --mixed-menus
run "AltObs 21x21" --raster alt
run "8x8 (Flat id default)" --size 8
run "12x12" --size 12
