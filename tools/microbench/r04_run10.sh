cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q --timeout 900 -k "perf_floors or bench_json" > gpurun_out/t9.log 2>&1; echo "pytest rc=$?" >> gpurun_out/t9.log; tail -4 gpurun_out/t9.log
for a in "" "--size 32" "--raster alt" "--size 8 --max-steps 100" "--size 12"; do CW_TUNE_VERBOSE=1 python bench.py --quick $a 2>&1 >/dev/null | grep craftingworld; done
bash tools/microbench/r04_shapes.sh
