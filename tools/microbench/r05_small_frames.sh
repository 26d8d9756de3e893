#!/bin/bash
# GPU box: the sweep of SMALL frames (under 4 KiB: Ray grids up to 9x9, AltObs up to 11x11) -- the product's rule (four workgroups per CU, the gather
# painter cw_render_gather_kernel for Ray grids up to 7x7) against the round-4 painter (cw_render_pieces_kernel, one workgroup per CU) and the two
# ingredients on their own.  (The third painter of profiles/r05_experiments.txt D, the piece staged in LDS, was measured from a working tree and removed.)
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
run product CW_X=0
run round4 CW_TUNE_GATHER=0 CW_TUNE_SMALL_BLOCKS=1
run pieces_4wg CW_TUNE_GATHER=0
[ "$RASTERS" = ray ] && run gather_1wg CW_TUNE_GATHER_MAX_SIZE=9 CW_TUNE_SMALL_BLOCKS=1
[ "$RASTERS" = ray ] && run gather_4wg CW_TUNE_GATHER_MAX_SIZE=9
