#!/bin/bash
# GPU box, from the repo root:   bash tools/microbench/run_exp.sh <spec file> [repetitions]
# One parametrised runner for the A/B experiments behind profiles/*.txt (it replaced rounds 1-2's exp_*.sh, one script per experiment).
# A spec file holds one arm per line:
#     label | VAR=value VAR=value ... | bench.py arguments
# ('#' starts a comment).  The arms run in order, the whole list `repetitions` times (alternating arms on ONE box is the only
# comparison that means anything: boxes differ by a few per cent).  An arm whose label starts with "pytest" runs the GPU test suite
# with the given environment instead (its third field goes to pytest, e.g. -k pixel).  Every arm is a fresh process.
# Output: one line per arm on stdout and in gpurun_out/<spec name>.txt.
SPEC=$1; REPS=${2:-2}
[ -f "$SPEC" ] || { echo "usage: $0 <spec file> [repetitions]"; exit 2; }
OUT=gpurun_out/$(basename "$SPEC" .spec).txt; mkdir -p gpurun_out; : > "$OUT"
fmt='import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; w=d.get("metric_window") or {}
print("%-46s %.4e env-steps/s  %.4f ms/step  kernel avg %.4f median %.4f ms (min %.4f) frac %.3f / %.3f%s" % (sys.argv[1], d["value"], d["ms_per_step"],
      r["avg_launch_ms"], r["median_launch_ms"] or 0, r["launch_ms_min_max"][0], r["frac"], r.get("frac_at_median_launch") or 0,
      ("  window %.4e" % w["value"]) if w else ""))'
python bench.py --quick --steps 300 --warmup 20 >/dev/null 2>&1      # the first process on a fresh box pages the image in: discarded
for rep in $(seq 1 "$REPS"); do
  echo "-- repetition $rep" | tee -a "$OUT"
  while IFS='|' read -r label envs args; do
    label=$(echo "$label" | sed 's/^ *//;s/ *$//'); [ -z "$label" ] && continue; case "$label" in \#*) continue;; esac
    if [ "${label#pytest}" != "$label" ]; then
      [ "$rep" = 1 ] || continue
      env $envs timeout -k 10 900 python -m pytest tests -m gpu -q $args > gpurun_out/last_pytest.txt 2>&1
      tail -40 gpurun_out/last_pytest.txt >> "${OUT%.txt}_pytest.txt"; line=$(tail -1 gpurun_out/last_pytest.txt)
      printf '%-46s %s\n' "$label" "$line" | tee -a "$OUT"
      continue
    fi
    env $envs timeout -k 10 600 python bench.py --quick --steps 600 --warmup 20 $args 2>gpurun_out/last_arm.err | python -c "$fmt" "$label" | tee -a "$OUT"
    grep -h "craftingworld\] placement\|craftingworld\] per-step render" gpurun_out/last_arm.err | cut -c1-260 | sed 's/^/      /' | tee -a "$OUT"
  done < "$SPEC"
done
true
