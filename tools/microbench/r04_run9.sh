cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q --timeout 900 > gpurun_out/t8.log 2>&1; echo "pytest rc=$?" >> gpurun_out/t8.log; tail -6 gpurun_out/t8.log
for a in "" "--size 32" "--raster alt" "--size 8 --max-steps 100" "--envs-per-gpu 1048576"; do CW_TUNE_VERBOSE=1 python bench.py --quick $a 2>&1 >/dev/null | grep craftingworld; done
python bench.py > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err; tail -c 600 gpurun_out/bench_full.err
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rm -rf gpurun_out/prof3 && rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof3 -o p3 -- python bench.py --quick --steps 600 > /dev/null 2>&1
