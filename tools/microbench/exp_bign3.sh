# GPU box: where should the linear sweep hand over to the frame-per-wave painter?  (2560 rounds per wave = ~374 000 envs at 21x21 so far)
run() { python bench.py --quick --steps 100 --warmup 10 "${@:2}" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-44s %.4e env-steps/s  %.4f ms/step  %s avg %.4f median %.4f ms frac %.3f' % ('$1', d['value'], d['ms_per_step'], r['kernel'], r['avg_launch_ms'], r['median_launch_ms'], r['frac']))"; true; }
for n in 393216 524288 786432 1048576; do
  for rep in 1 2; do
    CW_TUNE_RENDER_LINEAR=2 run "$n envs, linear sweep (forced)" --envs-per-gpu $n
    CW_TUNE_RENDER_LINEAR=2 CW_TUNE_RENDER_PACE=257 run "$n envs, linear sweep, m+1" --envs-per-gpu $n
    CW_TUNE_RENDER_LINEAR=0 run "$n envs, frame per wave" --envs-per-gpu $n
  done
done
