# GPU box: records-in-lanes render variant, in-situ (bench) and standalone, over render block counts
for b in 0 128 160 192 224; do
  echo "render blocks=$b (0 = default 256)"
  CW_TUNE_RENDER_BLOCKS=$b python tools/microbench/time_render.py preceding 2>/dev/null | head -1
  CW_TUNE_RENDER_BLOCKS=$b python bench.py --no-cpu-baseline --no-other-modes 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  bench value %.3e ms/step %.4f render %.4f' % (d['value'], d['ms_per_step'], d['kernels_ms']['render']))"
done
