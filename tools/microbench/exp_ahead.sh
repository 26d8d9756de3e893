for d in 1 2 4; do for b in 128 192 256; do
  echo "ahead=$d blocks=$b"
  CW_LIB_PATH=$PWD/gym_craftingworld_amd/libcw_ahead$d.so CW_TUNE_RENDER_BLOCKS=$b python tools/microbench/time_render.py preceding 2>/dev/null | head -1
  CW_LIB_PATH=$PWD/gym_craftingworld_amd/libcw_ahead$d.so CW_TUNE_RENDER_BLOCKS=$b python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  bench value %.3e ms/step %.4f render %.4f reset %.4f' % (d['value'], d['ms_per_step'], d['kernels_ms']['render'], d['kernels_ms']['reset']))"
done; done
