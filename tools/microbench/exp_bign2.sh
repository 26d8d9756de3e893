# GPU box: batches between the headline's and the switch to the frame-per-wave painter (2560 rounds per wave): base pace m+0 / m+1 / m+2, linear sweep vs frame per wave
run() { python bench.py --quick --steps 200 --warmup 20 "${@:2}" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-44s %.4e env-steps/s  %.4f ms/step  %s avg %.4f median %.4f ms frac %.3f' % ('$1', d['value'], d['ms_per_step'], r['kernel'], r['avg_launch_ms'], r['median_launch_ms'], r['frac']))"; true; }
run "warm-up (discard)"
for n in 131072 196608 262144 327680; do
  for rep in 1 2; do
    for p in 256 257 258; do CW_TUNE_RENDER_PACE=$p run "$n envs, linear, pace $p" --envs-per-gpu $n; done
    CW_TUNE_RENDER_LINEAR=0 run "$n envs, frame per wave" --envs-per-gpu $n
  done
done
