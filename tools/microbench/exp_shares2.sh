# GPU box: do the XCD shares (calibrated at cw_create) still pay once the sweep is paced?  linear and flat sweeps, shares on / off, alternating.
run() { python bench.py --quick --steps 300 "${@:2}" 2>gpurun_out/sh_err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1  value %.4e ms/step %.4f render %.4f (min %.4f max %.4f) frac %.3f' % (d['value'], d['ms_per_step'], d['kernels_ms']['render'] or 0, d['roofline']['launch_ms_min_max'][0], d['roofline']['launch_ms_min_max'][1], d['roofline']['frac']))"; true; }
run "warm-up (discard)               "
for rep in 1 2 3; do
  CW_TUNE_RENDER_PACE=256 run "linear m+0, shares              "
  CW_TUNE_RENDER_PACE=256 CW_TUNE_RENDER_SHARES=0 run "linear m+0, equal               "
  CW_TUNE_RENDER_PACE=257 run "linear m+1, shares              "
  CW_TUNE_RENDER_PACE=257 CW_TUNE_RENDER_SHARES=0 run "linear m+1, equal               "
  run "linear tuner, shares            "
  CW_TUNE_RENDER_SHARES=0 run "linear tuner, equal             "
  CW_TUNE_RENDER_FLAT=1 CW_TUNE_RENDER_FLAT_BLOCKS_PER_CU=8 CW_TUNE_RENDER_PACE=0 CW_TUNE_RENDER_SHARES=0 run "flat8 pace 0, equal             "
  CW_TUNE_RENDER_FLAT=1 CW_TUNE_RENDER_FLAT_BLOCKS_PER_CU=4 CW_TUNE_RENDER_PACE=0 CW_TUNE_RENDER_SHARES=0 run "flat4 pace 0, equal             "
done
