# GPU box: linear sweep vs frame per wave at large batches (frames resident: 3 x N x 21 168 B)
run() { python bench.py --quick --steps 100 --warmup 10 "${@:2}" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-44s %.4e env-steps/s  %.4f ms/step  render %.4f ms  frac %.3f' % ('$1', d['value'], d['ms_per_step'], r['avg_launch_ms'], r['frac']))"; true; }
for n in 262144 524288 1048576; do
  CW_TUNE_RENDER_ADAPT=0 run "$n envs, linear (calibrated)" --envs-per-gpu $n
  CW_TUNE_RENDER_PACE=257 run "$n envs, linear m+1" --envs-per-gpu $n
  CW_TUNE_RENDER_PACE=259 run "$n envs, linear m+3" --envs-per-gpu $n
  CW_TUNE_RENDER_LINEAR=0 run "$n envs, frame per wave" --envs-per-gpu $n
done
