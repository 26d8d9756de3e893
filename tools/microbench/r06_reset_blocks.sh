#!/bin/bash
# GPU box: workgroups per CU of the resetting kernels (cw_refill_kernel's scan, cw_reset_kernel): the short-episode scenario (refills of ~30 000 envs every 64
# steps), an explicit reset() of the batch, and the headline window (two all-env time-outs: a refill of 65 536 envs each).   bash tools/microbench/r06_reset_blocks.sh
cd ${GRAFT_REPO_ROOT:-.}
for b in 2 4 8; do
  echo "== CW_TUNE_RESET_BLOCKS=$b"
  CW_TUNE_RESET_BLOCKS=$b python tools/microbench/r06_short_episodes.py state 8 2>&1 | grep "period default" | head -1
  CW_TUNE_RESET_BLOCKS=$b python tools/microbench/time_reset.py 2>&1 | tail -1
  CW_TUNE_RESET_BLOCKS=$b python bench.py --quick --steps 600 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline K=600', d['value'], d['ms_per_step'])"
done
