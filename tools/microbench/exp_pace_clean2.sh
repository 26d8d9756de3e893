# GPU box: base pace m+0 (default), extra sleeps only beside >= 32 resetting waves, tuned online (default) vs fixed; several shapes
run() { python bench.py --quick --steps ${STEPS:-600} --warmup 20 "${@:2}" 2>gpurun_out/tuner_err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-44s %.4e env-steps/s  %.4f ms/step  kernel avg %.4f median %.4f ms (min %.4f) frac %.3f' % ('$1', d['value'], d['ms_per_step'], r['avg_launch_ms'], r['median_launch_ms'], r['launch_ms_min_max'][0], r['frac']))"; grep online gpurun_out/tuner_err.txt | tail -1 | cut -c1-150; true; }
export CW_TUNE_VERBOSE=1
run "warm-up (discard)"
for rep in 1 2; do
  run "sync,   default"
  for b in 0 1 2 3 4; do
    CW_TUNE_RENDER_PACE_BESIDE=$b run "desync, fixed +$b" --desync
  done
  run "desync, tuner (default)" --desync
  STEPS=2400 run "desync, tuner, 2400 steps" --desync
  run "131072 mixed menus, default" --envs-per-gpu 131072 --mixed-menus
  run "131072 mixed menus desync, tuner" --envs-per-gpu 131072 --mixed-menus --desync
  run "32x32, default" --size 32
  run "32x32 desync, tuner" --size 32 --desync
done
