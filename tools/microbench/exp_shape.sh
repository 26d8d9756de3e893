# GPU box: store shapes of the linear-sweep render kernel (CW_TUNE_RENDER_SHAPE = 0 lane = cell, 4 x 12 B at the cell's pixel rows |
# 3 contiguous 12-B chunks per lane | 4 contiguous 16-B chunks per lane), parity first (render / fixture tests under each shape), then
# forced paces and the calibrated + online-tuned default, alternating on one box.
run() { python bench.py --quick --steps 300 "${@:2}" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1  value %.4e ms/step %.4f render %.4f (min %.4f max %.4f) frac %.3f' % (d['value'], d['ms_per_step'], d['kernels_ms']['render'] or 0, d['roofline']['launch_ms_min_max'][0], d['roofline']['launch_ms_min_max'][1], d['roofline']['frac']))"; true; }
for shape in 3 4; do
  CW_TUNE_RENDER_SHAPE=$shape timeout -k 10 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "golden or fixture or render or random or storm or facade" > gpurun_out/shape_parity_$shape.txt 2>&1 || { tail -30 gpurun_out/shape_parity_$shape.txt; exit 1; }
  echo "shape $shape parity: $(tail -1 gpurun_out/shape_parity_$shape.txt)"
done
run "warm-up (discard)          "
for rep in 1 2; do
  for shape in 0 3 4; do
    export CW_TUNE_RENDER_SHAPE=$shape
    run "shape $shape, calibrated + tuner "
    for pace in 0 2 4 256 257 258 260; do
      CW_TUNE_RENDER_PACE=$pace run "shape $shape, pace $pace          "
    done
  done
done
