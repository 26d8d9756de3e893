# GPU box: the FLAT sweep render (CW_TUNE_RENDER_FLAT=1: 128-B-aligned 3072-B jobs over the frame array as one byte stream) vs the linear
# sweep (jobs = runs of grid rows): parity first (the GPU suite's frame tests under the flat sweep), then alternating bench runs.
run() { python bench.py --quick --steps 300 "${@:2}" 2>gpurun_out/flat_err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1  value %.4e ms/step %.4f render %.4f (min %.4f max %.4f) frac %.3f' % (d['value'], d['ms_per_step'], d['kernels_ms']['render'] or 0, d['roofline']['launch_ms_min_max'][0], d['roofline']['launch_ms_min_max'][1], d['roofline']['frac']))"; grep craftingworld gpurun_out/flat_err.txt; true; }
if [ "$1" != "noparity" ]; then
  CW_TUNE_RENDER_FLAT=1 timeout -k 10 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q > gpurun_out/flat_parity.txt 2>&1 || { tail -40 gpurun_out/flat_parity.txt; exit 1; }
  echo "flat sweep parity: $(tail -1 gpurun_out/flat_parity.txt)"
fi
export CW_TUNE_VERBOSE=1
run "warm-up (discard)               "
for rep in 1 2; do
  run "linear, calibrated + tuner      "
  CW_TUNE_RENDER_PACE=256 run "linear, pace m+0                "
  for bpc in 1 2 4 6 8; do
    CW_TUNE_RENDER_FLAT=1 CW_TUNE_RENDER_FLAT_BLOCKS_PER_CU=$bpc run "flat, $bpc blocks/CU, calibrated    "
    CW_TUNE_RENDER_FLAT=1 CW_TUNE_RENDER_FLAT_BLOCKS_PER_CU=$bpc CW_TUNE_RENDER_PACE=0 run "flat, $bpc blocks/CU, pace 0        "
  done
done
