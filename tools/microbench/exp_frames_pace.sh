# GPU box: does the frame-per-wave kernel want pacing too?  (it paints grids wider than 64 cells and batches above 2 560 rounds per wave)
run() { python bench.py --quick "${@:2}" 2>gpurun_out/fp_err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-50s %.4e env-steps/s  %.4f ms/step  render %.4f ms  frac %.3f' % ('$1', d['value'], d['ms_per_step'], r['avg_launch_ms'], r['frac']))"; grep "render pace" gpurun_out/fp_err.txt | head -2; true; }
export CW_TUNE_VERBOSE=1
CW_TUNE_RENDER_PACE=0 run "2^20 envs, frame per wave, unpaced" --envs-per-gpu 1048576 --steps 100 --warmup 10
run "2^20 envs, frame per wave, calibrated pace" --envs-per-gpu 1048576 --steps 100 --warmup 10
CW_TUNE_RENDER_PACE=256 run "2^20 envs, frame per wave, m+0" --envs-per-gpu 1048576 --steps 100 --warmup 10
CW_TUNE_RENDER_PACE=0 run "64x64, unpaced" --size 64 --steps 100 --warmup 10
run "64x64, calibrated pace" --size 64 --steps 100 --warmup 10
CW_TUNE_RENDER_PACE=256 run "64x64, m+0" --size 64 --steps 100 --warmup 10
CW_TUNE_RENDER_LINEAR=0 CW_TUNE_RENDER_PACE=0 run "65536 envs 21x21, frame per wave, unpaced" --steps 300
CW_TUNE_RENDER_LINEAR=0 run "65536 envs 21x21, frame per wave, calibrated" --steps 300
CW_TUNE_RENDER_LINEAR=0 CW_TUNE_RENDER_PACE=256 run "65536 envs 21x21, frame per wave, m+0" --steps 300
run "65536 envs 21x21, linear (default)" --steps 300
