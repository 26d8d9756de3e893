#!/bin/bash
# GPU box only: HBM traffic of the bench's kernels from the PMC counters, collected as
# MI355X_MICROARCH.md prescribes: separate --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one
# pass), no tracing domains besides --kernel-trace, plus a calibration pass on a kernel with a
# known byte count (tools/microbench/store_variants).
#   bash tools/profile_pmc.sh <tag> [bench.py arguments of the shape, e.g. --size 32 | --desync]   -> gpurun_out/pmc_<tag>/summary.json
set -e -o pipefail
export TMPDIR=/tmp
TAG=${1:-headline}; shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
# the calibration kernel is built from its source here (no binary is committed)
make -B -C tools/microbench store_variants > $OUT/build_calib.log 2>&1
for C in WRITE_SIZE FETCH_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/bench_$C -- python bench.py --quick --steps 60 --warmup 5 "$@" > $OUT/bench_$C.json 2> $OUT/bench_$C.err
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/calib_$C -- ./tools/microbench/store_variants > $OUT/calib_$C.txt 2> $OUT/calib_$C.err
done
python tools/summarize_pmc.py $OUT > $OUT/summary.json
cat $OUT/summary.json
