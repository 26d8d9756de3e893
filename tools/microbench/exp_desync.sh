# GPU box: full-frame mode with the episode phases spread out (~218 resets beside the render kernel on every step): who gets the SIMD?
run() { python bench.py --quick --steps 600 "${@:2}" 2>gpurun_out/desync_err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-58s %.4e env-steps/s  %.4f ms/step  render %.4f ms (min %.4f max %.4f)' % ('$1', d['value'], d['ms_per_step'], r['avg_launch_ms'], r['launch_ms_min_max'][0], r['launch_ms_min_max'][1]))"; true; }
for rep in 1 2; do
for prio in 1 0 2; do
  for pace in 256 257 258; do
    CW_TUNE_RESET_PRIO=$prio CW_TUNE_RENDER_PACE=$pace run "desync: prio mode $prio (1 reset high, 0 none, 2 render high), m+$((pace-256))" --desync
  done
done
done
CW_TUNE_RESET_PRIO=2 CW_TUNE_RENDER_PACE=256 run "sync: prio mode 2, m+0"
CW_TUNE_RESET_PRIO=1 CW_TUNE_RENDER_PACE=256 run "sync: prio mode 1, m+0"
