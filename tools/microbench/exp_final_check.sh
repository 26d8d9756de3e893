run() { python bench.py --quick --steps ${STEPS:-600} --warmup 20 "${@:2}" 2>gpurun_out/tuner_err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-44s %.4e env-steps/s  %.4f ms/step  kernel avg %.4f median %.4f ms (min %.4f) frac %.3f' % ('$1', d['value'], d['ms_per_step'], r['avg_launch_ms'], r['median_launch_ms'], r['launch_ms_min_max'][0], r['frac']))"; grep online gpurun_out/tuner_err.txt | tail -1 | cut -c1-150; true; }
export CW_TUNE_VERBOSE=1
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
run "warm-up (discard)"
for rep in 1 2; do
  run "sync,   default"
  CW_TUNE_RENDER_ADAPT=0 run "sync,   tuner off"
  run "desync, tuner (default)" --desync
  CW_TUNE_RENDER_PACE_BESIDE=1 run "desync, fixed +1" --desync
  STEPS=2400 run "desync, tuner, 2400 steps" --desync
done
