#!/bin/bash
# GPU box only: where do the sweep's cycles go (cw_render_pieces_kernel)?  One --pmc pass per
# counter group (SQ: 8 slots, TCC: 4 slots per pass), kernel-trace only beside it.   usage: bash tools/profile_counters.sh [tag] [bench.py arguments, e.g. --desync]
set -e -o pipefail
export TMPDIR=/tmp
TAG=${1:-sync}; shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/counters_$TAG
rm -rf $OUT; mkdir -p $OUT
i=0
for G in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_WR" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_IFETCH SQ_INSTS_SMEM" \
         "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum" \
         "TCC_BUSY_sum TCC_TAG_STALL_sum TCC_WRITE_sum TCC_WRITEBACK_sum" \
         "TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_LATENCY_sum TA_TA_BUSY_sum" \
         "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $G --kernel-trace --output-format csv -d $OUT/g$i -- python bench.py --quick --steps 60 --warmup 5 "$@" > $OUT/g$i.json 2> $OUT/g$i.err || echo "group $i failed"
done
OUT=$OUT python - <<'PY'
import csv, glob, os, collections, json
out = os.environ['OUT']
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '').split('<')[0]
        if k.startswith('cw_'):
            acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
res = {k: {c: sorted(v)[len(v) // 2] for c, v in d.items()} for k, d in acc.items()}
json.dump(res, open(out + '/summary.json', 'w'), indent=1)
print(json.dumps({k: v for k, v in res.items() if 'render' in k}, indent=1))
PY
