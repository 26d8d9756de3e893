#!/bin/bash
# GPU box: variant (b) of the step kernel after the sweep -- more, narrower waves (CW_TUNE_STEP_ENVS_PER_WAVE) and s_setprio(3) (make exp EXP=-DCW_EXP_STEP_PRIO=1 NAME=stepprio)
export TMPDIR=/tmp
cd /tmp; cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/step_b; mkdir -p $O
run() { label=$1; shift
  for mode in sync desync; do
    extra=""; [ $mode = desync ] && extra="--desync"
    rm -rf $O/rp_${label}_$mode
    env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $O/rp_${label}_$mode -o p -- python bench.py --quick --steps 600 $extra > $O/${label}_${mode}_rp.json 2> $O/${label}_${mode}_rp.err
    python - <<PY
import csv, glob
f = sorted(glob.glob('$O/rp_${label}_$mode/**/p_kernel_stats.csv', recursive=True))[0]
k = {r['Name'].split('(')[0].replace('void ', ''): r for r in csv.DictReader(open(f))}
st, sw = k['cw_step_fused_kernel'], [v for n, v in k.items() if n.startswith('cw_render_pieces_kernel')][0]
print('%-10s %-6s step kernel avg %.2f us (min %.2f), sweep avg %.2f us' % ('$label', '$mode', float(st['AverageNs']) / 1e3, float(st['MinNs']) / 1e3, float(sw['AverageNs']) / 1e3), flush=True)
PY
  done
}
run epw64 CW_X=0
run epw32 CW_TUNE_STEP_ENVS_PER_WAVE=32
run epw16 CW_TUNE_STEP_ENVS_PER_WAVE=16
run prio CW_LIB_PATH=$PWD/gym_craftingworld_amd/libcw_exp_stepprio.so
