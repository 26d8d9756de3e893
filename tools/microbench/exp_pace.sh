# GPU box: pacing sweep of the linear-sweep render kernel (CW_TUNE_RENDER_PACE = n sleeps of 64 clocks per pair of jobs, +256 = one more
# inside each job), online tuner off, one box.  (The LDS-staged and timing-only variants of profiles/r02_render_linear.txt sections I-J
# were removed from the source after they had been measured.)
run() { python bench.py --quick --steps 300 "${@:2}" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1  value %.4e ms/step %.4f render %.4f (min %.4f max %.4f)' % (d['value'], d['ms_per_step'], d['kernels_ms']['render'] or 0, d['roofline']['launch_ms_min_max'][0], d['roofline']['launch_ms_min_max'][1]))"; true; }
run "warm-up (discard)   "
CW_TUNE_RENDER_LINEAR=0 run "frame per wave, calibrated pace"
for pace in 0 1 2 3 4 6 8; do
  CW_TUNE_RENDER_PACE=$pace run "linear, pace $pace      "
done
for pace in 0 1 2 3 4; do
  CW_TUNE_RENDER_PACE=$((256 + pace)) run "linear, pace m+$pace    "
done
run "linear, calibrated + online tuner"
