# GPU box: the one-launch step with the frame-per-wave painter (cw_render_frames_step_kernel) vs two kernels on two streams: wide grids, large batches, AltObs
run() { python bench.py --quick --warmup 20 "${@:2}" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-46s %.4e env-steps/s  %.4f ms/step  %s %.4f ms (min %.4f max %.4f) frac %.3f' % ('$1', d['value'], d['ms_per_step'], r['kernel'], r['avg_launch_ms'], r['launch_ms_min_max'][0], r['launch_ms_min_max'][1], r['frac']))"; true; }
timeout -k 10 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "launch_arrangements or golden or altobs or alt" > gpurun_out/fused5_parity.txt 2>&1 || { tail -30 gpurun_out/fused5_parity.txt; exit 1; }
echo "parity: $(tail -1 gpurun_out/fused5_parity.txt)"
run "warm-up (discard)" --steps 300
for rep in 1 2; do
  run "64x64, one launch" --size 64 --steps 100
  CW_TUNE_FUSED_RENDER=0 run "64x64, two kernels" --size 64 --steps 100
  run "2^20 envs, one launch" --envs-per-gpu 1048576 --steps 100
  CW_TUNE_FUSED_RENDER=0 run "2^20 envs, two kernels" --envs-per-gpu 1048576 --steps 100
  run "AltObs, one launch" --raster alt --steps 600
  CW_TUNE_FUSED_RENDER=0 run "AltObs, two kernels" --raster alt --steps 600
  run "AltObs desync, one launch" --raster alt --steps 600 --desync
  CW_TUNE_FUSED_RENDER=0 run "AltObs desync, two kernels" --raster alt --steps 600 --desync
done
