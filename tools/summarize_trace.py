#!/usr/bin/env python3
"""Per-kernel launch times of a `rocprofv3 --kernel-trace` run of bench.py over the TIMED launches only.

rocprofv3's own --stats table averages every launch of the process: cw_create's calibration of the sweep's clock (~260 sweeps at seven rates) and the
warm-up steps are in it.  bench.py --quick takes prewarm + W untimed steps, K timed steps (`value`) and K more with the library's HIP events around the
dominant kernel (`roofline.avg_launch_ms`): this reads the per-launch trace and reports the dominant kernel's and the step kernel's averages over exactly
those two K-launch regions -- the figure to hold against roofline.avg_launch_ms.

    python tools/summarize_trace.py <rocprofv3 output dir> <bench line .json>   -> JSON on stdout
"""
import csv
import glob
import json
import os
import sys


def main(rp_dir, bench_json):
    d = json.loads(open(bench_json).read().strip().splitlines()[-1])
    K = d['steps']
    f = sorted(glob.glob(os.path.join(rp_dir, '**', '*_kernel_trace.csv'), recursive=True))[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
    by = {}
    for r in rows:
        name = r['Kernel_Name'].split('(')[0].replace('void ', '')
        by.setdefault(name, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    dom = d['roofline']['kernel_in_trace']
    cand = [n for n in by if n.replace(' ', '') == dom.replace(' ', '')] or [n for n in by if n.startswith(d['roofline']['kernel'])]
    out = {'trace': os.path.relpath(f), 'steps_per_region': K, 'kernels': {}}
    launches_per_step = {}
    for name in cand + [n for n in by if n.startswith('cw_step_fused_kernel') and n not in cand]:
        us = by[name]
        per_step = max(1, round(len(us) / max(1, (d['warmup_total'] + 2 * K + 260))))     # (a chunked sweep: several launches per step)
        launches_per_step[name] = per_step
        n = K * per_step
        if len(us) < 2 * n:
            continue
        timed, profiled = us[-2 * n:-n], us[-n:]
        out['kernels'][name] = {
            'launches_in_trace': len(us), 'launches_per_step': per_step,
            'timed_region_avg_us': sum(timed) / len(timed) * per_step, 'timed_region_median_us': sorted(timed)[len(timed) // 2] * per_step,
            'profiled_region_avg_us': sum(profiled) / len(profiled) * per_step,
            'all_launches_avg_us': sum(us) / len(us)}
    dk = out['kernels'].get(cand[0]) if cand else None
    if dk:
        out['against_bench_line'] = {'roofline.avg_launch_ms': d['roofline']['avg_launch_ms'], 'trace_profiled_region_ms': dk['profiled_region_avg_us'] / 1e3,
                                     'ratio': dk['profiled_region_avg_us'] / 1e3 / d['roofline']['avg_launch_ms'],
                                     'note': 'the same launches: the library\'s HIP events on the launch stream against rocprofv3\'s kernel timestamps'}
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
