#!/bin/bash
# GPU box: bench.py --quick on the shapes that are not the headline (default tuning) -> one line each
cd ${GRAFT_REPO_ROOT:-.}
run() { label=$1; shift
  python bench.py --quick --steps 600 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; t=d['tuner']
print('%-34s %.4e env-steps/s  ms/step %.4f  kernel avg %.4f med %.4f min %.4f  frac %.3f / %.3f  period %.0f ns' % ('$label', d['value'], d['ms_per_step'], r['avg_launch_ms'], r['median_launch_ms'], r['launch_ms_min_max'][0], r['frac'], r['frac_at_median_launch'], t['period16'] / 1.6))"
}
run "65536 21x21" 
run "65536 21x21 desync" --desync
run "131072 21x21 mixed menus" --envs-per-gpu 131072 --mixed-menus
run "131072 21x21 mixed desync" --envs-per-gpu 131072 --mixed-menus --desync
run "65536 32x32" --size 32
run "65536 32x32 desync" --size 32 --desync
run "65536 alt 21x21" --raster alt
run "65536 alt 21x21 desync" --raster alt --desync
run "65536 alt 32x32" --raster alt --size 32
run "65536 12x12" --size 12
run "65536 8x8 (max_steps 100)" --size 8 --max-steps 100
run "65536 8x8 desync" --size 8 --max-steps 100 --desync
run "262144 8x8" --size 8 --max-steps 100 --envs-per-gpu 262144
run "65536 5x5" --size 5 --max-steps 100
run "262144 5x5" --size 5 --max-steps 100 --envs-per-gpu 262144
run "16384 64x64" --size 64 --envs-per-gpu 16384
run "1048576 21x21 (8 chunks)" --envs-per-gpu 1048576
