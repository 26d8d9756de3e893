#!/bin/bash
# GPU box: the dirty-cell step, product against throw-away builds (make -C gym_craftingworld_amd/csrc exp EXP=-DCW_EXP_DIRTY=n NAME=dirtyn):
#   1 no stores at all, 2 one dword per changed env, 3 the same number of 12-byte stores but all inside the frame's first two lines
cd ${GRAFT_REPO_ROOT:-.}
for lib in product "$@"; do
  for mode in sync desync; do
    extra=""; [ $mode = desync ] && extra="--desync"
    if [ $lib = product ]; then unset CW_LIB_PATH; else export CW_LIB_PATH=$PWD/gym_craftingworld_amd/libcw_exp_$lib.so; fi
    python bench.py --quick --steps 2000 --obs-mode pixels_dirty $extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-10s dirty $mode  %.4e env-steps/s  us/step %.2f  kernel us %.2f' % ('$lib', d['value'], d['ms_per_step'] * 1e3, d['kernels_ms']['step'] * 1e3))"
  done
done
