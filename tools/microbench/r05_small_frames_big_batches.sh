#!/bin/bash
# GPU box: frames under 4 KiB in LARGER batches -- how many workgroups per CU (CW_TUNE_SMALL_BLOCKS) when the sweep is no longer launch-bound?
cd ${GRAFT_REPO_ROOT:-.}
for n in 131072 262144; do for s in 5 7 8 9; do for wg in 1 2 4; do
  CW_TUNE_SMALL_BLOCKS=$wg python bench.py --quick --steps 300 --size $s --envs-per-gpu $n 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%7d envs %dx%d  %d wg/CU  %4.0f MB  sweep %.4f ms (median %.4f) frac %.3f  period16 %d  %s' % ($n, $s, $s, $wg, r['algorithmic_bytes_per_launch'] / 1e6, r['avg_launch_ms'], r['median_launch_ms'], r['frac'], d['tuner']['period16'], r['kernel_in_trace']))"
done; done; done
