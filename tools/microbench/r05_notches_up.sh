cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2; do for n in "0.4 0.75" "0.4 0.9" "0.4 1.1" "0.4 1.4" "0.55 0.9" "0.55 1.1"; do set -- $n
  for mode in desync sync; do
    extra=""; [ $mode = desync ] && extra="--desync"
    CW_TUNE_PERIOD_NS=545 CW_TUNE_HEAD_NOTCH=$1 CW_TUNE_BUSY_NOTCH=$2 python bench.py --quick --steps 600 $extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('head %s busy %s %-6s value %.4e  ms/step %.4f  sweep %.4f ms (median %.4f, max %.4f) frac %.3f' % ('$1', '$2', '$mode', d['value'], d['ms_per_step'], r['avg_launch_ms'], r['median_launch_ms'], r['launch_ms_min_max'][1], r['frac']))"
  done
done; done
