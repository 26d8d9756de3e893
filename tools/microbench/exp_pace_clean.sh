# GPU box: with the extra sleeps applied only beside >= 32 resetting waves: base pace once more (phases in step), and the extra (phases spread out)
run() { python bench.py --quick --steps ${STEPS:-600} --warmup 20 "${@:2}" 2>gpurun_out/tuner_err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-40s %.4e env-steps/s  %.4f ms/step  kernel avg %.4f median %.4f ms (min %.4f) frac %.3f' % ('$1', d['value'], d['ms_per_step'], r['avg_launch_ms'], r['median_launch_ms'], r['launch_ms_min_max'][0], r['frac']))"; true; }
run "warm-up (discard)"
for rep in 1 2; do
  for p in 0 1 2 256 257 258; do
    CW_TUNE_RENDER_PACE=$p run "sync,   base pace $p"
  done
  for b in 0 2 3 4 5 6; do
    CW_TUNE_RENDER_PACE_BESIDE=$b run "desync, base 0, fixed +$b" --desync
  done
  run "desync, base 0, tuner" --desync
  STEPS=2400 run "desync, base 0, tuner, 2400 steps" --desync
done
