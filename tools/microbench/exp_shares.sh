# GPU box: XCD-aware frame shares (calibrated at cw_create) vs equal shares, alternating on one box
run() { python bench.py --no-cpu-baseline --no-other-modes 2>gpurun_out/shares_err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1  value %.4e ms/step %.4f render %.4f' % (d['value'], d['ms_per_step'], d['kernels_ms']['render'] or 0))"; grep craftingworld gpurun_out/shares_err.txt; true; }
export CW_TUNE_VERBOSE=1
run "warm-up (discard)  "
for rep in 1 2 3; do
  CW_TUNE_RENDER_SHARES=0 run "equal shares       " || exit 1
  run "calibrated shares  " || exit 1
done
