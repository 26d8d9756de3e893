# GPU box: what cw_create's calibration (launches queued back to back) picks, in several processes, against forced paces, same box
run() { python bench.py --quick --steps 300 "${@:2}" 2>gpurun_out/calib_err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1  value %.4e ms/step %.4f render %.4f (min %.4f max %.4f) frac %.3f' % (d['value'], d['ms_per_step'], d['kernels_ms']['render'] or 0, d['roofline']['launch_ms_min_max'][0], d['roofline']['launch_ms_min_max'][1], d['roofline']['frac']))"; grep craftingworld gpurun_out/calib_err.txt | cut -c1-260; true; }
export CW_TUNE_VERBOSE=1
run "warm-up (discard)        "
for rep in 1 2 3 4; do
  run "calibrated               "
  CW_TUNE_RENDER_SHARES=0 run "calibrated, equal shares "
  CW_TUNE_RENDER_PACE=257 run "forced m+1               "
done
