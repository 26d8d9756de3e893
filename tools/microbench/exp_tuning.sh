run() { echo "== $*"; env "$@" python bench.py --no-cpu-baseline --steps 600 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  value %.3e ms/step %.4f render %.4f reset %.4f step %.4f' % (d['value'], d['ms_per_step'], d['kernels_ms']['render'], d['kernels_ms']['reset'], d['kernels_ms']['step']))"; }
run CW_TUNE_OVERLAP=1
run CW_TUNE_OVERLAP=0
run CW_TUNE_RESET_PRIO=0
run CW_TUNE_RENDER_BLOCKS_PER_CU=4
run CW_TUNE_RENDER_BLOCKS_PER_CU=8
run CW_TUNE_LIST_BLOCKS=32
run CW_TUNE_LIST_BLOCKS=32 CW_TUNE_RENDER_BLOCKS_PER_CU=4
run CW_TUNE_OVERLAP=1
