# GPU box: one-launch step: base pace m+0 / m+1 x extra sleeps while envs are reset beside the sweep (CW_TUNE_RENDER_PACE_BESIDE), alternating on one box
run() { python bench.py --quick --steps 600 --warmup 20 "${@:2}" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-46s %.4e env-steps/s  %.4f ms/step  %s %.4f ms (min %.4f max %.4f) frac %.3f' % ('$1', d['value'], d['ms_per_step'], r['kernel'], r['avg_launch_ms'], r['launch_ms_min_max'][0], r['launch_ms_min_max'][1], r['frac']))"; true; }
run "warm-up (discard)"
for rep in 1 2 3; do
  CW_TUNE_RENDER_PACE=256 run "sync,   m+0"
  CW_TUNE_RENDER_PACE=257 run "sync,   m+1"
  for b in 1 2 3 4; do
    CW_TUNE_RENDER_PACE=256 CW_TUNE_RENDER_PACE_BESIDE=$b run "desync, m+0, +$b beside resets" --desync
    CW_TUNE_RENDER_PACE=257 CW_TUNE_RENDER_PACE_BESIDE=$b run "desync, m+1, +$b beside resets" --desync
  done
done
