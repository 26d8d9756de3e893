# GPU box: phases spread out (resets beside every render launch): forced paces (+1 is added in the kernel while envs are being reset)
run() { python bench.py --quick --steps 300 --warmup 20 "${@:2}" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-52s %.4e env-steps/s  %.4f ms/step  %s %.4f ms  frac %.3f' % ('$1', d['value'], d['ms_per_step'], r['kernel'], r['avg_launch_ms'], r['frac']))"; true; }
run "warm-up (discard)"
for rep in 1 2; do
  run "sync, default"
  run "desync, default (m+1, +1 beside resets)" --desync
  for p in 256 257 258; do CW_TUNE_RENDER_PACE=$p run "desync, pace $p (+1 beside resets)" --desync; done
  run "131072 mixed menus desync, default" --envs-per-gpu 131072 --mixed-menus --desync
  CW_TUNE_RENDER_PACE=258 run "131072 mixed menus desync, pace 258" --envs-per-gpu 131072 --mixed-menus --desync
  run "32x32 desync, default" --size 32 --desync
done
