#!/bin/bash
# GPU box: the step kernel that follows the sweep (cw_step_fused_kernel, full-frame mode), product against throw-away builds
#   (make -C gym_craftingworld_amd/csrc exp EXP=... NAME=...): plain runs for the rate, rocprofv3 --kernel-trace --stats for the kernels' own times.
#   bash tools/microbench/r05_step_ab.sh [lib names, e.g. nocoop touch]     -> stdout
export TMPDIR=/tmp
cd /tmp; cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/step_ab; mkdir -p $O
for lib in product "$@"; do
  if [ $lib = product ]; then unset CW_LIB_PATH; else export CW_LIB_PATH=$PWD/gym_craftingworld_amd/libcw_exp_$lib.so; fi
  for mode in sync desync; do
    extra=""; [ $mode = desync ] && extra="--desync"
    python bench.py --quick --steps 600 $extra > $O/${lib}_$mode.json 2>/dev/null
    rm -rf $O/rp_${lib}_$mode
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/rp_${lib}_$mode -o p -- python bench.py --quick --steps 600 $extra > $O/${lib}_${mode}_rp.json 2> $O/${lib}_${mode}_rp.err
    python - <<PY
import json, csv, glob
d = json.loads(open('$O/${lib}_$mode.json').read().strip().splitlines()[-1])
f = sorted(glob.glob('$O/rp_${lib}_$mode/**/p_kernel_stats.csv', recursive=True))[0]
k = {r['Name'].split('(')[0].replace('void ', ''): r for r in csv.DictReader(open(f))}
st, sw = k['cw_step_fused_kernel'], [v for n, v in k.items() if n.startswith('cw_render_pieces_kernel')][0]
print('%-13s %-6s value %.4e  ms/step %.4f  sweep(ev) %.4f frac %.3f step_frac %.3f | rocprof: step kernel avg %.2f us (min %.2f, max %.1f), sweep avg %.2f us' % (
    '$lib', '$mode', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline']['step_frac'],
    float(st['AverageNs']) / 1e3, float(st['MinNs']) / 1e3, float(st['MaxNs']) / 1e3, float(sw['AverageNs']) / 1e3), flush=True)
PY
  done
done
