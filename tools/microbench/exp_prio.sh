run() { echo "== $*"; env "$@" python bench.py --no-cpu-baseline --steps 600 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  value %.3e ms/step %.4f render %.4f reset %.4f step %.4f' % (d['value'], d['ms_per_step'], d['kernels_ms']['render'], d['kernels_ms']['reset'], d['kernels_ms']['step']))"; }
for i in 1 2 3; do
run CW_TUNE_RESET_PRIO=0
run CW_TUNE_RESET_PRIO=1
done
run CW_TUNE_RESET_PRIO=0 CW_TUNE_OVERLAP=0
run CW_TUNE_RESET_PRIO=1 CW_TUNE_OVERLAP=0
