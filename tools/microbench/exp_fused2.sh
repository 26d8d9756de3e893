# GPU box: the one-launch step (cw_render_step_kernel): paces and wave priorities once more, phases in step and spread out, alternating on one box
run() { python bench.py --quick --steps 600 --warmup 20 "${@:2}" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-46s %.4e env-steps/s  %.4f ms/step  %s %.4f ms (min %.4f max %.4f) frac %.3f' % ('$1', d['value'], d['ms_per_step'], r['kernel'], r['avg_launch_ms'], r['launch_ms_min_max'][0], r['launch_ms_min_max'][1], r['frac']))"; true; }
run "warm-up (discard)"
for rep in 1 2; do
  for p in 256 257 258; do
    CW_TUNE_RENDER_PACE=$p run "sync,   pace $p"
    CW_TUNE_RENDER_PACE=$p run "desync, pace $p (+1 beside resets)" --desync
  done
  for prio in 0 1 2; do
    CW_TUNE_RESET_PRIO=$prio run "sync,   priorities mode $prio"
    CW_TUNE_RESET_PRIO=$prio run "desync, priorities mode $prio" --desync
  done
done
