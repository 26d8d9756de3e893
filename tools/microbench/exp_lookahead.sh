# GPU box: look-ahead resets (CW_TUNE_LOOKAHEAD=1) vs inline resets, state-only and dirty-cell modes, synchronized and spread-out
# episodes, eager launches and HIP graphs of 16 steps
run() { python bench.py --quick --obs-mode $2 --steps 608 --warmup 16 ${@:3} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 %-13s %-28s value %.4e  %.2f us/step' % ('$2', '${*:3}', d['value'], d['ms_per_step']*1e3))"; true; }
for mode in state pixels_dirty; do
  for args in "" "--desync" "--graph-steps 16" "--graph-steps 16 --desync"; do
    CW_TUNE_LOOKAHEAD=0 run "inline    " $mode $args
    CW_TUNE_LOOKAHEAD=1 run "look-ahead" $mode $args
  done
done
