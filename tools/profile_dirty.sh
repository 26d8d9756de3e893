#!/bin/bash
# GPU box only: what bounds the step kernel of the reference's own observation strategy (render_edit, ray.py:522-557 = obs_mode pixels_dirty)?
# cw_step_fused_kernel with paint == 1 against the same kernel in the state-only mode: rocprofv3 kernel times, then one --pmc pass per counter
# group (TCC: 4 slots, FETCH_SIZE 3 / WRITE_SIZE 2; SQ: 8; separate passes, kernel-trace only beside them).
#   bash tools/profile_dirty.sh [modes, default "pixels_dirty state"] [extra bench.py arguments, e.g. --desync]   -> gpurun_out/dirty/summary.json
set -o pipefail
export TMPDIR=/tmp
cd /tmp; cd $GRAFT_REPO_ROOT
MODES=${1:-"pixels_dirty state"}; shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/dirty
rm -rf $OUT; mkdir -p $OUT
for mode in $MODES; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$mode -o p -- python bench.py --quick --steps 600 --obs-mode $mode "$@" > $OUT/bench_stats_$mode.json 2> $OUT/stats_$mode.err
  cp $(ls $OUT/stats_$mode/p_kernel_stats.csv $OUT/stats_$mode/*/p_kernel_stats.csv 2>/dev/null | head -1) $OUT/kernel_stats_$mode.csv
  i=0
  for G in "WRITE_SIZE" "FETCH_SIZE" \
           "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
           "TCC_REQ_sum TCC_WRITE_sum TCC_READ_sum TCC_WRITEBACK_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum" \
           "TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_LATENCY_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_VALU" \
           "GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    rocprofv3 --pmc $G --kernel-trace --output-format csv -d $OUT/${mode}_g$i -- python bench.py --quick --steps 120 --warmup 5 --obs-mode $mode "$@" > $OUT/${mode}_g$i.json 2> $OUT/${mode}_g$i.err || echo "$mode group $i failed"
    echo "$mode group $i done"
  done
done
OUT=$OUT MODES="$MODES" python - <<'PY'
import csv, glob, os, collections, json
out, modes = os.environ['OUT'], os.environ['MODES'].split()
res = {'note': 'median per launch of cw_step_fused_kernel over a bench.py --quick run (65 536 envs, 21x21); FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports '
               'them (x 1024 -> bytes; gfx950: FETCH_SIZE counts half of wide coalesced reads); *_sum counters summed over the 8 XCDs'}
for mode in modes:
    acc = collections.defaultdict(list)
    for f in glob.glob(out + '/%s_g*/**/*counter_collection.csv' % mode, recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Kernel_Name'].split('(')[0].replace('void ', '').split('<')[0] == 'cw_step_fused_kernel':
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
    d = {c: sorted(v)[len(v) // 2] for c, v in acc.items()}
    d['launches_sampled'] = max((len(v) for v in acc.values()), default=0)
    for r in csv.DictReader(open(out + '/kernel_stats_%s.csv' % mode)):
        if r['Name'].startswith('cw_step_fused_kernel'):
            d['kernel_avg_ns'], d['kernel_min_ns'], d['kernel_max_ns'], d['kernel_calls'] = float(r['AverageNs']), float(r['MinNs']), float(r['MaxNs']), int(r['Calls'])
    try:
        d['bench'] = {k: v for k, v in json.loads(open(out + '/bench_stats_%s.json' % mode).read().strip().splitlines()[-1]).items() if k in ('value', 'ms_per_step', 'config')}
    except Exception:  # noqa: BLE001
        pass
    res[mode] = d
json.dump(res, open(out + '/summary.json', 'w'), indent=1)
print(json.dumps(res, indent=1))
PY
