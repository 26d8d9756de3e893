# GPU box: why is the flat sweep slower inside a step sequence than in cw_create's back-to-back calibration launches?
run() { python bench.py --quick --steps 300 "${@:2}" 2>gpurun_out/flat_err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1  value %.4e ms/step %.4f render %.4f (min %.4f max %.4f) frac %.3f' % (d['value'], d['ms_per_step'], d['kernels_ms']['render'] or 0, d['roofline']['launch_ms_min_max'][0], d['roofline']['launch_ms_min_max'][1], d['roofline']['frac']))"; grep "with shares" gpurun_out/flat_err.txt; true; }
export CW_TUNE_VERBOSE=1
run "warm-up (discard)               "
CW_TUNE_RENDER_PACE=256 run "linear, pace m+0                "
export CW_TUNE_RENDER_FLAT=1 CW_TUNE_RENDER_FLAT_BLOCKS_PER_CU=8
run "flat8, calibrated + tuner       "
CW_TUNE_RENDER_SHARES=0 run "flat8, equal shares             "
CW_TUNE_RESET_PRIO=0 run "flat8, no priorities            "
CW_TUNE_RESET_PRIO=1 run "flat8, reset waves raised       "
CW_TUNE_OVERLAP=0 run "flat8, no side stream           "
for pace in 0 1 2 4 256 258; do
  CW_TUNE_RENDER_PACE=$pace run "flat8, pace $pace                  "
done
CW_TUNE_RENDER_SHARES=0 CW_TUNE_RENDER_PACE=0 run "flat8, pace 0, equal shares     "
python tools/microbench/time_render.py rate pixels
CW_TUNE_RENDER_FLAT=0 python tools/microbench/time_render.py rate pixels
