# GPU box: on a box where unpaced is the best pace (the sweep looks issue-bound): more sweep waves?
run() { python bench.py --quick --steps 600 --warmup 20 "${@:2}" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-34s %.4e env-steps/s  %.4f ms/step  kernel avg %.4f median %.4f ms (min %.4f) frac %.3f  fill %.0f GB/s' % ('$1', d['value'], d['ms_per_step'], r['avg_launch_ms'], r['median_launch_ms'], r['launch_ms_min_max'][0], r['frac'], r['fill_same_bytes_GBs']))"; true; }
python tools/microbench/clock_trace.py 0 0.3 2>/dev/null | grep -E "^t 0.2" | head -2 | cut -c1-260
run "warm-up (discard)"
for rep in 1 2; do
  CW_TUNE_RENDER_PACE=0 run "1 workgroup/CU, pace 0"
  CW_TUNE_RENDER_PACE=256 run "1 workgroup/CU, pace m+0"
  for p in 0 256 2 4 258 260; do
    CW_TUNE_RENDER_BLOCKS_PER_CU=2 CW_TUNE_RENDER_PACE=$p run "2 workgroups/CU, pace $p"
  done
done
