# GPU box: FULL pixel step, render + auto-resets as ONE launch (cw_render_step_kernel, default) vs two kernels on two streams (CW_TUNE_FUSED_RENDER=0)
# vs one stream, resets before the render (CW_TUNE_OVERLAP=0); parity first, then alternating bench runs, phases in step and spread out.
run() { python bench.py --quick --steps 600 --warmup 20 "${@:2}" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-46s %.4e env-steps/s  %.4f ms/step  %s %.4f ms (min %.4f max %.4f) frac %.3f' % ('$1', d['value'], d['ms_per_step'], r['kernel'], r['avg_launch_ms'], r['launch_ms_min_max'][0], r['launch_ms_min_max'][1], r['frac']))"; true; }
if [ "$1" != "noparity" ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/fused_parity.txt 2>&1 || { tail -40 gpurun_out/fused_parity.txt; exit 1; }
  echo "GPU suite with the fused launch: $(tail -1 gpurun_out/fused_parity.txt)"
fi
run "warm-up (discard)"
for rep in 1 2 3; do
  run "sync,   one launch"
  CW_TUNE_FUSED_RENDER=0 run "sync,   two kernels, two streams"
  CW_TUNE_OVERLAP=0 run "sync,   one stream, resets first"
  run "desync, one launch" --desync
  CW_TUNE_FUSED_RENDER=0 run "desync, two kernels, two streams" --desync
done
run "131072 mixed menus, one launch" --envs-per-gpu 131072 --mixed-menus
CW_TUNE_FUSED_RENDER=0 run "131072 mixed menus, two kernels" --envs-per-gpu 131072 --mixed-menus
run "32x32, one launch" --size 32
CW_TUNE_FUSED_RENDER=0 run "32x32, two kernels" --size 32
