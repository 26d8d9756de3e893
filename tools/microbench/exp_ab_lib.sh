# GPU box: A/B two builds of the library on the same box, alternating.  usage: exp_ab_lib.sh <other.so> [bench args]
OTHER=$1; shift
for rep in 1 2 3; do
  for lib in default $OTHER; do
    if [ $lib = default ]; then unset CW_LIB_PATH; else export CW_LIB_PATH=$PWD/$lib; fi
    python bench.py --no-cpu-baseline --no-other-modes "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib  value %.4e ms/step %.4f render %.4f' % (d['value'], d['ms_per_step'], d['kernels_ms']['render'] or 0))"
  done
done
