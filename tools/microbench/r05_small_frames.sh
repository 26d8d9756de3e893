#!/bin/bash
# GPU box: the sweep of SMALL frames, the round-4 painter (cw_render_pieces_kernel: fill, then the items on top) against
#   staged  -- the piece put together in LDS and stored once (CW_TUNE_STAGED=1), at 1 / 2 / 4 workgroups per CU (CW_TUNE_GATHER_BLOCKS)
#   gather  -- every lane computes its own 16-byte chunks (CW_TUNE_GATHER=1, Ray grids up to 9x9)
#   bash tools/microbench/r05_small_frames.sh "sizes" "rasters" [envs]
cd ${GRAFT_REPO_ROOT:-.}
SIZES=${1:-"4 5 6 7 8 9 12"}; RASTERS=${2:-"ray"}; N=${3:-65536}
run() { # label, env assignments...
  label=$1; shift
  for raster in $RASTERS; do for s in $SIZES; do
    env "$@" python bench.py --quick --steps 600 --size $s --raster $raster --envs-per-gpu $N 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-12s %-3s %2dx%-2d %7d envs  value %.4e  ms/step %.4f  sweep %.4f ms (median %.4f)  frac %.3f (median %.3f)  period16 %d  %s' % ('$label', '$raster', $s, $s, $N, d['value'], d['ms_per_step'], r['avg_launch_ms'], r['median_launch_ms'], r['frac'], r['frac_at_median_launch'], d['tuner']['period16'], r['kernel_in_trace']))"
  done; done
}
run round4 CW_X=0
run staged_1wg CW_TUNE_STAGED=1
run staged_2wg CW_TUNE_STAGED=1 CW_TUNE_GATHER_BLOCKS=2
run staged_4wg CW_TUNE_STAGED=1 CW_TUNE_GATHER_BLOCKS=4
[ "$RASTERS" = ray ] && run gather_4wg CW_TUNE_GATHER=1 CW_TUNE_GATHER_BLOCKS=4
