# GPU box: the render kernel's variants alternating on one box: frame per wave (CW_TUNE_RENDER_LINEAR=0) | linear sweep, pace
# calibrated at cw_create (default) | linear sweep with forced pace (256 + n = with the sleep inside each job)
run() { python bench.py --no-cpu-baseline --no-other-modes --no-single-env --steps 300 "${@:2}" 2>gpurun_out/linear_err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1  value %.4e ms/step %.4f render %.4f (min %.4f max %.4f) window %.4e' % (d['value'], d['ms_per_step'], d['kernels_ms']['render'] or 0, d['roofline']['launch_ms_min_max'][0], d['roofline']['launch_ms_min_max'][1], d['metric_window']['value']))"; grep craftingworld gpurun_out/linear_err.txt; true; }
export CW_TUNE_VERBOSE=1
run "warm-up (discard)                 "
for rep in 1 2 3; do
  CW_TUNE_RENDER_LINEAR=0 run "frame per wave                    " "$@" || exit 1
  run "linear, calibrated pace + shares  " "$@" || exit 1
  CW_TUNE_RENDER_PACE=0 run "linear, pace 0                    " "$@" || exit 1
  CW_TUNE_RENDER_PACE=257 run "linear, pace m+1                  " "$@" || exit 1
done
