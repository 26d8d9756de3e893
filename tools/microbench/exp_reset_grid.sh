# GPU box: resetting workgroups per CU (the all-env reset step of the synchronized benchmark; phases spread out), alternating on one box
run() { python bench.py --no-cpu-baseline --no-other-modes --no-single-env --steps 600 --warmup 20 "${@:2}" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; w=d['metric_window']; print('%-40s %.4e env-steps/s  %.4f ms/step  kernel %.4f ms (max %.4f)  window %.4e slowest steps %s' % ('$1', d['value'], d['ms_per_step'], r['avg_launch_ms'], r['launch_ms_min_max'][1], w['value'], w['slowest_step_ms']))"; true; }
for rep in 1 2; do
  for b in 1 2 4; do
    CW_TUNE_FUSED_RESET_BLOCKS_PER_CU=$b run "sync,   $b resetting workgroups per CU"
  done
  for b in 1 2 4; do
    CW_TUNE_FUSED_RESET_BLOCKS_PER_CU=$b run "desync, $b resetting workgroups per CU" --desync
  done
done
