# GPU box: pacing the linear render's stores (CW_TUNE_RENDER_PACE = s_sleep(1), 64 clocks, per pair of jobs), three store paths
run() { python bench.py --no-cpu-baseline --no-other-modes --no-single-env --steps 300 "${@:2}" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1  value %.4e ms/step %.4f render %.4f (min %.4f max %.4f)' % (d['value'], d['ms_per_step'], d['kernels_ms']['render'] or 0, d['roofline']['launch_ms_min_max'][0], d['roofline']['launch_ms_min_max'][1]))"; true; }
run "warm-up (discard)   "
CW_TUNE_RENDER_LINEAR=0 run "frame per wave        "
for pace in 0 4 8 10 12 14; do
  CW_TUNE_RENDER_PACE=$pace CW_TUNE_RENDER_LINEAR=1 run "1 LDS 16 B,    pace $pace "
done
for pace in 0 2 4 6 8 10; do
  CW_TUNE_RENDER_PACE=$pace CW_TUNE_RENDER_LINEAR=2 run "2 direct 12 B, pace $pace "
done
for pace in 0 2 4 6 8; do
  CW_TUNE_RENDER_PACE=$pace CW_TUNE_RENDER_LINEAR=3 run "3 TIMING ONLY, pace $pace "
done
CW_TUNE_RENDER_LINEAR=0 run "frame per wave        "
