# GPU box: the pinned placement of render_groups' batch loop (.p2align 5 + 3 x s_nop 0, the working tree's library) against the last unpinned build
# (libcw_head.so, built from the commit before) -- and how to look for a neighbour: build variants with k = 0..7 s_nop after the .p2align
# (-DCW_EXP_PAD ... as in tools/microbench/exp_pad.sh) and alternate them here.
run() { python bench.py --quick --steps 600 --warmup 20 "${@:2}" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-34s %.4e env-steps/s  %.4f ms/step  kernel avg %.4f median %.4f ms (min %.4f) frac %.3f' % ('$1', d['value'], d['ms_per_step'], r['avg_launch_ms'], r['median_launch_ms'], r['launch_ms_min_max'][0], r['frac']))"; true; }
D=$PWD/gym_craftingworld_amd
run "warm-up (discard)"
for rep in 1 2 3; do
  CW_LIB_PATH=$D/libcw_head.so run "unpinned (previous commit)"
  run "pinned"
  CW_LIB_PATH=$D/libcw_head.so run "unpinned, desync" --desync
  run "pinned, desync" --desync
done
run "pinned, 131072 mixed menus" --envs-per-gpu 131072 --mixed-menus
run "pinned, 32x32" --size 32
