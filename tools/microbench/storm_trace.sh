#!/bin/bash
# GPU box: storm_step.py under rocprofv3 --kernel-trace: the durations of the step kernel and of the sweep on and around the all-env time-out steps
export TMPDIR=/tmp; cd /tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/storm_trace
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/storm_trace -o s -- python tools/microbench/storm_step.py > gpurun_out/storm_trace.txt 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/storm_trace/**/s_kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
seq = [(r['Kernel_Name'][:24], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3) for r in rows if 'cw_' in r['Kernel_Name']]
# the longest step kernels = the time-out steps; print them with the sweep that follows and the neighbours
idx = sorted(range(len(seq)), key=lambda i: -seq[i][1] if 'step_fused' in seq[i][0] else 0)[:2]
for i in sorted(idx):
    print(' | '.join('%s %.1f us' % seq[j] for j in range(max(0, i - 2), min(len(seq), i + 5))))
PY
