#!/bin/bash
# GPU box: the full-frame step as ONE launch, workgroup by workgroup (CW_TUNE_FUSED=1: cw_step_sweep_kernel) against the product's two (cw_step_fused_kernel,
# then the sweep): plain runs for the rate, two repetitions, arms alternating.
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2; do for fused in 0 1; do for mode in sync desync; do
  extra=""; [ $mode = desync ] && extra="--desync"
  CW_TUNE_FUSED=$fused python bench.py --quick --steps 600 $extra "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('fused=$fused %-6s value %.4e  ms/step %.4f  bracketed kernel %s %.4f ms (median %.4f) frac %.3f  step_frac %.3f  period16 %d guard %d' % ('$mode', d['value'], d['ms_per_step'], r['kernel'], r['avg_launch_ms'], r['median_launch_ms'], r['frac'], r['step_frac'], d['tuner']['period16'], d['tuner']['guard_slowdowns']))"
done; done; done
