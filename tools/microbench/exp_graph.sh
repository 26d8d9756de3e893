run() { echo "== $*"; python bench.py --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  value %.3e ms/step %.4f' % (d['value'], d['ms_per_step']))"; }
run --obs-mode state --envs-per-gpu 4096 --steps 1200 --warmup 16
run --obs-mode state --envs-per-gpu 4096 --steps 1200 --warmup 16 --graph-steps 16
run --obs-mode state --steps 1200 --warmup 16
run --obs-mode state --steps 1200 --warmup 16 --graph-steps 16
run --obs-mode pixels_dirty --steps 1200 --warmup 16
run --obs-mode pixels_dirty --steps 1200 --warmup 16 --graph-steps 16
