# GPU box: alternate two builds on time_steps.py.  usage: exp_ab_steps.sh <other.so> <mode> <n> <sync|desync>
OTHER=$1; shift
for rep in 1 2 3; do
  for lib in default $OTHER; do
    if [ $lib = default ]; then unset CW_LIB_PATH; else export CW_LIB_PATH=$PWD/$lib; fi
    echo -n "$lib: "; python tools/microbench/time_steps.py "$@" 2>/dev/null
  done
done
