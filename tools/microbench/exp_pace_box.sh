# GPU box: full pace sweep of the one-launch step on whatever box this lands on (slow boxes: 0.247-0.249 ms per launch at m+0, fast ones 0.234-0.239)
run() { python bench.py --quick --steps 600 --warmup 20 "${@:2}" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-24s %.4e env-steps/s  %.4f ms/step  kernel avg %.4f median %.4f ms (min %.4f) frac %.3f  fill %.0f GB/s' % ('$1', d['value'], d['ms_per_step'], r['avg_launch_ms'], r['median_launch_ms'], r['launch_ms_min_max'][0], r['frac'], r['fill_same_bytes_GBs']))"; true; }
run "warm-up (discard)"
for rep in 1 2; do
  for p in 256 0 1 2 3 4 6 257 258 260; do
    CW_TUNE_RENDER_PACE=$p run "pace $p"
  done
done
