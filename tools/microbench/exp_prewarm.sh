# GPU box: bench.py with a driver-like short timed region (K=20, W=5) after 0 / 100 / 300 / 600 / 1500 untimed device warm-up steps; and the default K=600
run() { python bench.py --no-cpu-baseline --no-other-modes --no-single-env "${@:2}" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-40s value %.4e  repeats %s  window %.4e  kernel avg %.4f' % ('$1', d['value'], ' '.join('%.4e' % v for v in d['repeats']['value']), d['metric_window']['value'], r['avg_launch_ms']))"; true; }
for rep in 1 2; do
  for p in 0 100 300 600 1500; do
    run "K=20 W=5, $p device warm-up steps" --steps 20 --warmup 5 --prewarm-steps $p
  done
  run "K=600 W=20, no device warm-up" --prewarm-steps 0
  run "K=600 W=20, 600 (default)"
done
