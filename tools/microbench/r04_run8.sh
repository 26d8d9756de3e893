cd $GRAFT_REPO_ROOT
run() { label=$1; shift
  python bench.py --quick --steps 600 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; t=d['tuner']
print('%-34s %.4e env-steps/s  ms/step %.4f  kernel avg %.4f med %.4f min %.4f  frac %.3f / %.3f  period %.0f ns' % ('$label', d['value'], d['ms_per_step'], r['avg_launch_ms'], r['median_launch_ms'], r['launch_ms_min_max'][0], r['frac'], r['frac_at_median_launch'], t['period16'] / 1.6))"
}
for b in 1 2; do export CW_TUNE_SWEEP_BLOCKS_PER_CU=$b; echo "== blocks per CU $b"
run "65536 8x8" --size 8 --max-steps 100
run "262144 8x8" --size 8 --max-steps 100 --envs-per-gpu 262144
run "65536 5x5" --size 5 --max-steps 100
run "262144 5x5" --size 5 --max-steps 100 --envs-per-gpu 262144
run "65536 alt 21x21" --raster alt
run "65536 9x9" --size 9
done
unset CW_TUNE_SWEEP_BLOCKS_PER_CU
for s in 21 32; do CW_TUNE_VERBOSE=1 python bench.py --quick --size $s 2>&1 >/dev/null | grep craftingworld; done
CW_TUNE_VERBOSE=1 python bench.py --quick --raster alt 2>&1 >/dev/null | grep craftingworld
run "65536 21x21" 
run "65536 21x21 desync" --desync
run "65536 32x32" --size 32
run "1048576 21x21 (8 chunks)" --envs-per-gpu 1048576
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof2 -o p2 -- python bench.py --quick --steps 600 --desync > /dev/null 2>&1
