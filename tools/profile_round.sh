#!/bin/bash
# GPU box only: the round's evidence set -> gpurun_out/round/ (copy what is judged into profiles/).
#   1. bench.py default run (value, roofline from HIP events, cpu_baseline)
#   2. rocprofv3 --kernel-trace --stats of the same command
#   3. PMC traffic passes + calibration (tools/profile_pmc.sh)
set -e -o pipefail
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/round
rm -rf $OUT; mkdir -p $OUT
python bench.py > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rocprof -- python bench.py --quick > $OUT/bench_under_rocprof.json 2> $OUT/rocprof.err
cp $OUT/rocprof/*/*kernel_stats.csv $OUT/kernel_stats.csv
bash tools/profile_pmc.sh > $OUT/pmc.log 2>&1
cp $GRAFT_REPO_ROOT/gpurun_out/pmc/summary.json $OUT/pmc_traffic.json
cat $OUT/bench.json; cut -c1-160 $OUT/kernel_stats.csv | head -8
