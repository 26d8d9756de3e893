run() { echo "== $*"; python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  value %.3e ms/step %.4f kernels %s' % (d['value'], d['ms_per_step'], d['kernels_ms']))"; }
run --obs-mode state --envs-per-gpu 4096 --steps 1200
run --obs-mode state --envs-per-gpu 65536 --steps 1200
run --obs-mode state --envs-per-gpu 1048576 --steps 1200
run --obs-mode pixels_dirty --envs-per-gpu 65536 --steps 1200
run --obs-mode pixels --envs-per-gpu 65536 --size 32 --steps 600
run --obs-mode pixels --envs-per-gpu 131072 --steps 600
