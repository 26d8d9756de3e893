#!/bin/bash
# GPU box: cw_step_fused_kernel specialised on <PAINT, TERM> (round 6) against the round-5 kernel (one body, `paint` a runtime argument):
#   tools builds the old one as gym_craftingworld_amd/libcw_exp_r05step.so (git show f5ad428:.../cw_kernels.hip with today's cw_engine.cpp).
#   the kernels' own times from rocprofv3 --kernel-trace --stats: bench.py --quick --steps 3000 --obs-mode state / pixels_dirty (the step kernel IS the
#   step there), and --steps 600 in the full-frame mode (the step kernel that FOLLOWS the sweep); episode phases in step and spread out.
#   bash tools/microbench/r06_step_variants.sh   -> stdout
export TMPDIR=/tmp
cd /tmp; cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r06_step; mkdir -p $O
for lib in product r05step; do
  if [ $lib = product ]; then unset CW_LIB_PATH; else export CW_LIB_PATH=$PWD/gym_craftingworld_amd/libcw_exp_$lib.so; fi
  for obs in state pixels_dirty; do
    for mode in sync desync; do
      extra=""; [ $mode = desync ] && extra="--desync"
      rm -rf $O/rp_${lib}_${obs}_$mode
      rocprofv3 --kernel-trace --stats --output-format csv -d $O/rp_${lib}_${obs}_$mode -o p -- python bench.py --quick --steps 3000 --obs-mode $obs $extra > $O/${lib}_${obs}_${mode}_rp.json 2> $O/${lib}_${obs}_${mode}_rp.err
      python - <<PY
import json, csv, glob
d = json.loads(open('$O/${lib}_${obs}_${mode}_rp.json').read().strip().splitlines()[-1])
f = sorted(glob.glob('$O/rp_${lib}_${obs}_$mode/**/p_kernel_stats.csv', recursive=True))[0]
k = {r['Name'].split('(')[0].replace('void ', ''): r for r in csv.DictReader(open(f))}
st = [v for n, v in k.items() if n.startswith('cw_step_fused_kernel')][0]
rf = [v for n, v in k.items() if n.startswith('cw_refill_kernel')][0]
print('%-9s %-12s %-6s value (under rocprof) %.4e  us/step %.2f | rocprof: step kernel avg %.2f us (min %.2f, max %.1f, %s calls), refill avg %.2f us (%s calls)' % (
    '$lib', '$obs', '$mode', d['value'], d['ms_per_step'] * 1e3, float(st['AverageNs']) / 1e3, float(st['MinNs']) / 1e3, float(st['MaxNs']) / 1e3, st['Calls'],
    float(rf['AverageNs']) / 1e3, rf['Calls']), flush=True)
PY
    done
  done
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
st = [v for n, v in k.items() if n.startswith('cw_step_fused_kernel')][0]
sw = [v for n, v in k.items() if n.startswith('cw_render_pieces_kernel')][0]
print('%-9s full frames %-6s value %.4e  ms/step %.4f  sweep(ev) %.4f frac %.3f step_frac %.3f | rocprof: step kernel avg %.2f us (min %.2f, max %.1f), sweep avg %.2f us' % (
    '$lib', '$mode', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline']['step_frac'],
    float(st['AverageNs']) / 1e3, float(st['MinNs']) / 1e3, float(st['MaxNs']) / 1e3, float(sw['AverageNs']) / 1e3), flush=True)
PY
  done
done
