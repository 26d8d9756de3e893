# GPU box: one-launch step: is base pace m+0 stable (it was bistable on one box with the two-stream step)?  5 alternating runs each
run() { python bench.py --quick --steps 600 --warmup 20 "${@:2}" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-46s %.4e env-steps/s  %.4f ms/step  %s %.4f ms (min %.4f max %.4f) frac %.3f' % ('$1', d['value'], d['ms_per_step'], r['kernel'], r['avg_launch_ms'], r['launch_ms_min_max'][0], r['launch_ms_min_max'][1], r['frac']))"; true; }
run "warm-up (discard)"
for rep in 1 2 3 4 5; do
  CW_TUNE_RENDER_PACE=256 run "sync,   m+0"
  CW_TUNE_RENDER_PACE=257 run "sync,   m+1"
  CW_TUNE_RENDER_PACE=256 CW_TUNE_RENDER_PACE_BESIDE=3 run "desync, m+0, +3 beside resets" --desync
  CW_TUNE_RENDER_PACE=256 CW_TUNE_RENDER_PACE_BESIDE=2 run "desync, m+0, +2 beside resets" --desync
done
CW_TUNE_RENDER_PACE=256 run "131072 mixed menus, m+0" --envs-per-gpu 131072 --mixed-menus
CW_TUNE_RENDER_PACE=257 run "131072 mixed menus, m+1" --envs-per-gpu 131072 --mixed-menus
CW_TUNE_RENDER_PACE=256 run "32x32, m+0" --size 32
CW_TUNE_RENDER_PACE=257 run "32x32, m+1" --size 32
CW_TUNE_RENDER_PACE=256 run "262144, m+0" --envs-per-gpu 262144 --steps 200
CW_TUNE_RENDER_PACE=257 run "262144, m+1" --envs-per-gpu 262144 --steps 200
