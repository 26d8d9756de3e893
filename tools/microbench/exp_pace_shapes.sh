# GPU box: is the fixed pace m+1 right for the other shapes too?  forced paces vs the default vs the old calibration, alternating on one box
run() { python bench.py --quick --steps ${STEPS:-300} --warmup 20 "${@:2}" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-52s %.4e env-steps/s  %.4f ms/step  %s %.4f ms  frac %.3f' % ('$1', d['value'], d['ms_per_step'], r['kernel'], r['avg_launch_ms'], r['frac']))"; true; }
sweep() {   # label, bench args...
  run "$1, default (m+1 fixed)" "${@:2}"
  for p in 256 257 258 0 2; do CW_TUNE_RENDER_PACE=$p run "$1, pace $p" "${@:2}"; done
  CW_TUNE_RENDER_CALIBRATE=1 CW_TUNE_RENDER_SHARES=1 run "$1, calibrated + shares" "${@:2}"
  run "$1, default (m+1 fixed) again" "${@:2}"
}
run "warm-up (discard)"
sweep "21x21 65536"
sweep "21x21 65536 desync" --desync
sweep "21x21 131072 mixed menus" --envs-per-gpu 131072 --mixed-menus
sweep "32x32 65536" --size 32
sweep "8x8 65536" --size 8 --max-steps 100
STEPS=100 sweep "64x64 65536" --size 64
STEPS=100 sweep "21x21 2^20 (frame per wave)" --envs-per-gpu 1048576
