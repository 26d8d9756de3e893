# GPU box: cw_render alone, queued continuously, runs at 0.2235-0.2258 ms (m+1); inside cw_step the same kernel takes 0.2354.  What costs the 5 %?
run() { python bench.py --quick --steps 300 --warmup 20 "${@:2}" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-44s %.4e env-steps/s  %.4f ms/step  %s %.4f ms (min %.4f max %.4f) frac %.3f' % ('$1', d['value'], d['ms_per_step'], r['kernel'], r['avg_launch_ms'], r['launch_ms_min_max'][0], r['launch_ms_min_max'][1], r['frac']))"; true; }
run "warm-up (discard)"
for rep in 1 2; do
  run "default"
  CW_TUNE_OVERLAP=0 run "no side stream (reset after the render)"
  CW_TUNE_RESET_PRIO=0 run "no wave priorities"
  run "max_steps 60000 (no storm in the region)" --max-steps 60000
  run "HIP graphs of 16 steps" --graph-steps 16 --steps 320 --warmup 16
  python tools/microbench/clock_trace.py 257 1 2>/dev/null | grep -E "^==|^t 0\.9" | cut -c1-70
done
