#!/usr/bin/env python3
"""Summarise tools/profile_pmc.sh output into profiles/profiles/rNN_pmc_traffic.json content (stdout)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def kernel_base(name):
    """'void cw_render_step_kernel<3>(CwParams, int, ...)' -> 'cw_render_step_kernel' (the eight placements of the sweep loop are one kernel)"""
    name = name.split('(')[0].strip()
    if name.startswith('void '):
        name = name[5:]
    return name.split('<')[0]


def per_kernel(root, counter):
    files = glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True)
    acc = defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            if r.get('Counter_Name') == counter:
                acc[kernel_base(r['Kernel_Name'])].append(float(r['Counter_Value']))
    return acc


def main(out):
    res = {'unit_note': 'FETCH_SIZE/WRITE_SIZE are reported in KiB (x1024 -> bytes); gfx950: FETCH_SIZE counts 1/2 of '
                        'wide coalesced reads (MI355X_MICROARCH.md HBM section) -- raw and corrected values both listed'}
    calib = {}
    for c in ('WRITE_SIZE', 'FETCH_SIZE'):
        k = per_kernel(os.path.join(out, 'calib_' + c), c)
        for name, v in k.items():
            calib.setdefault(name, {})[c] = sum(v) / len(v) * 1024.0
    known = 65536 * 21168.0
    res['calibration_known_bytes_per_launch'] = known
    res['calibration'] = {n: {c: {'bytes': b, 'ratio_to_known': b / known} for c, b in d.items()} for n, d in calib.items()
                          if 'fill_x3' in n or 'fill_x4' in n or 'frame_cellrow' in n}
    kern = {}
    for c in ('WRITE_SIZE', 'FETCH_SIZE'):
        k = per_kernel(os.path.join(out, 'bench_' + c), c)
        for name, v in k.items():
            if name.startswith('cw_'):
                v2 = sorted(v)
                kern.setdefault(name, {})[c] = {'launches': len(v), 'mean_bytes': sum(v) / len(v) * 1024.0,
                                                'median_bytes': v2[len(v2) // 2] * 1024.0}
    res['kernels'] = kern
    # the kernel of the full-frame render: one launch with the auto-resets (default) or the linear sweep alone
    dom = next((k for k in ('cw_render_pieces_step_kernel', 'cw_render_step_kernel', 'cw_render_pieces_kernel') if k in kern), 'cw_render_kernel')
    r = kern.get(dom, {})
    if 'WRITE_SIZE' in r and 'FETCH_SIZE' in r:
        w, f = r['WRITE_SIZE']['median_bytes'], r['FETCH_SIZE']['median_bytes']
        res['hbm_bytes_per_launch'] = w + 2.0 * f
        res['hbm_bytes_per_launch_note'] = dom + ', median launch: WRITE_SIZE + 2 x FETCH_SIZE (gfx950 read correction)'
        res['algorithmic_bytes_per_launch'] = 65536 * (441 + 21168.0)
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    main(sys.argv[1])
