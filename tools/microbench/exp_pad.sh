# GPU box (EXPERIMENT): does the sweep's launch time depend on WHERE its inner loop lies in the code object?  k x s_nop 0 (4 bytes each, executed once per wave)
# inserted before the batch loop of render_groups (`Rec nxt = fetch(0);`); libraries alternate on one box (CW_LIB_PATH).  Build the variants with
#   for k in 1 2 3 4 6 8 11 13 16 24 32: hipcc ... -DCW_EXP_PAD=k "-DCW_EXP_PAD_ASM=<k x asm volatile(\"s_nop 0\");>" -o gym_craftingworld_amd/libcw_pad$k.so
# after putting `#ifdef CW_EXP_PAD / CW_EXP_PAD_ASM / #endif` at that line; libcw_pad0.so = the unpadded build.
run() { python bench.py --quick --steps 600 --warmup 20 "${@:2}" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-24s %.4e env-steps/s  %.4f ms/step  kernel avg %.4f median %.4f ms (min %.4f) frac %.3f' % ('$1', d['value'], d['ms_per_step'], r['avg_launch_ms'], r['median_launch_ms'], r['launch_ms_min_max'][0], r['frac']))"; true; }
D=$PWD/gym_craftingworld_amd
run "warm-up (discard)"
for rep in 1 2; do
  for k in 0 1 2 3 4 6 8 11 13 16 24 32; do
    CW_LIB_PATH=$D/libcw_pad$k.so run "pad $k x 4 bytes"
  done
done
