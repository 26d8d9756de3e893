#!/bin/bash
# GPU box only: the round's evidence set -> gpurun_out/round/ (copy what is judged into profiles/).
#   1. bench.py default run (value, roofline from HIP events, cpu_baseline)
#   2. rocprofv3 --kernel-trace --stats of bench.py --quick for every BASELINE shape: headline, phases spread out, 32x32, 131 072 mixed menus, AltObs
#   3. PMC traffic passes + calibration (tools/profile_pmc.sh) for the headline, the spread-out phases, 32x32, AltObs and 131 072 mixed menus
#   4. (consumer) rocprofv3 --kernel-trace --stats with a reader of every observation byte between two steps (bench.py --consumer reduce32)
#   bash tools/profile_round.sh [steps: bench | stats | pmc, default all]
set -e -o pipefail
export TMPDIR=/tmp
cd /tmp; cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/round
WHAT=${1:-"bench stats consumer pmc"}
mkdir -p $OUT
shape_args() { case $1 in headline) echo "";; desync) echo "--desync";; 32x32) echo "--size 32";; 131072_mixed) echo "--envs-per-gpu 131072 --mixed-menus";; alt) echo "--raster alt";; esac; }
for w in $WHAT; do case $w in
bench)
  python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 300 $OUT/bench.err;;
stats)
  for tag in headline desync 32x32 131072_mixed alt; do
    rm -rf $OUT/rocprof_$tag
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rocprof_$tag -o p -- python bench.py --quick $(shape_args $tag) > $OUT/bench_under_rocprof_$tag.json 2> $OUT/rocprof_$tag.err
    cp $(ls $OUT/rocprof_$tag/p_kernel_stats.csv $OUT/rocprof_$tag/*/p_kernel_stats.csv 2>/dev/null | head -1) $OUT/kernel_stats_$tag.csv
    # (the --stats table averages cw_create's calibration launches too: the per-launch trace, over the timed launches only)
    python tools/summarize_trace.py $OUT/rocprof_$tag $OUT/bench_under_rocprof_$tag.json > $OUT/kernel_trace_timed_$tag.json || echo "trace summary $tag failed"
    echo "== $tag"; cut -c1-150 $OUT/kernel_stats_$tag.csv | head -5; python -c "import json;print(json.load(open('$OUT/kernel_trace_timed_$tag.json')).get('against_bench_line'))"
  done;;
consumer)
  rm -rf $OUT/rocprof_consumer
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rocprof_consumer -o p -- python bench.py --quick --steps 300 --desync --consumer reduce32 > $OUT/bench_under_rocprof_consumer.json 2> $OUT/rocprof_consumer.err
  cp $(ls $OUT/rocprof_consumer/p_kernel_stats.csv $OUT/rocprof_consumer/*/p_kernel_stats.csv 2>/dev/null | head -1) $OUT/kernel_stats_consumer.csv
  echo "== consumer"; cut -c1-150 $OUT/kernel_stats_consumer.csv | head -5;;
pmc)
  for tag in headline desync 32x32 alt 131072_mixed; do
    bash tools/profile_pmc.sh $tag $(shape_args $tag) > $OUT/pmc_$tag.log 2>&1 || echo "pmc $tag failed"
    cp $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag/summary.json $OUT/pmc_traffic_$tag.json
    python -c "import json;d=json.load(open('$OUT/pmc_traffic_$tag.json'));print('$tag', d.get('hbm_bytes_per_launch'), d.get('algorithmic_bytes_per_launch'), d.get('traffic_over_algorithmic'))"
  done;;
esac; done
