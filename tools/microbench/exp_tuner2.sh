# GPU box: base pace 0 (default now); the online tuner on the extra sleeps beside resets (default) vs fixed values, phases in step and spread out
run() { python bench.py --quick --steps ${STEPS:-1200} --warmup 20 "${@:2}" 2>gpurun_out/tuner_err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-40s %.4e env-steps/s  %.4f ms/step  kernel avg %.4f median %.4f ms frac %.3f' % ('$1', d['value'], d['ms_per_step'], r['avg_launch_ms'], r['median_launch_ms'], r['frac']))"; grep -c "online" gpurun_out/tuner_err.txt | sed 's/^/      tuner moves: /'; grep online gpurun_out/tuner_err.txt | tail -2 | cut -c1-160; true; }
export CW_TUNE_VERBOSE=1
run "warm-up (discard)"
for rep in 1 2; do
  run "sync,   tuner (default)"
  CW_TUNE_RENDER_ADAPT=0 run "sync,   fixed +3"
  run "desync, tuner (default)" --desync
  for b in 2 3 5 6; do
    CW_TUNE_RENDER_PACE_BESIDE=$b run "desync, fixed +$b" --desync
  done
  STEPS=2400 run "desync, tuner, 2400 steps" --desync
done
