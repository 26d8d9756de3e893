# GPU box: online render-pace tuner (default) vs fixed paces, episodes in phase and spread out
run() { python bench.py --quick --steps 600 "${@:2}" 2>gpurun_out/adapt_err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-40s %.4e env-steps/s  %.4f ms/step  render %.4f ms (min %.4f max %.4f)' % ('$1', d['value'], d['ms_per_step'], r['avg_launch_ms'], r['launch_ms_min_max'][0], r['launch_ms_min_max'][1]))"; grep "online" gpurun_out/adapt_err.txt | tail -4; true; }
export CW_TUNE_VERBOSE=1
for rep in 1 2; do
  run "sync,   online tuner"
  CW_TUNE_RENDER_ADAPT=0 run "sync,   calibrated at create only"
  CW_TUNE_RENDER_PACE=257 run "sync,   fixed m+1"
  run "desync, online tuner" --desync
  CW_TUNE_RENDER_ADAPT=0 run "desync, calibrated at create only" --desync
  CW_TUNE_RENDER_PACE=257 run "desync, fixed m+1" --desync
done
run "2^20 envs, online tuner" --envs-per-gpu 1048576 --steps 200 --warmup 10
CW_TUNE_RENDER_ADAPT=0 run "2^20 envs, calibrated at create only" --envs-per-gpu 1048576 --steps 200 --warmup 10
CW_TUNE_RENDER_LINEAR=0 run "2^20 envs, frame per wave" --envs-per-gpu 1048576 --steps 200 --warmup 10
