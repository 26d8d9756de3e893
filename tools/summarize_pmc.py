#!/usr/bin/env python3
"""Summarise tools/profile_pmc.sh output into profiles/rNN_pmc_traffic_<tag>.json content (stdout): per kernel WRITE_SIZE / FETCH_SIZE per launch,
the calibration ratios, and for the dominant kernel (the sweep, cw_render_pieces_kernel) the HBM bytes per launch beside the algorithmic bytes of the
shape that ran (`shape`: what bench.py matches its own run against when it quotes `roofline.traffic`)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def kernel_base(name):
    """'void cw_render_pieces_kernel<0, 2>(CwParams, ...)' -> 'cw_render_pieces_kernel'"""
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
    # the shape that ran (the bench line of the WRITE_SIZE pass) and the sweep's traffic
    shape = None
    try:
        line = [l for l in open(os.path.join(out, 'bench_WRITE_SIZE.json')).read().strip().splitlines() if l.startswith('{')][-1]
        cfg = json.loads(line)['config']
        shape = {'envs_per_gpu': cfg['envs_per_gpu'], 'size': cfg['size'], 'obs_mode': cfg['obs_mode'], 'raster': cfg.get('raster', 'ray'),
                 'episode_phases': cfg['episode_phases'], 'task_lists': cfg['task_lists']}
    except Exception:  # noqa: BLE001
        pass
    res['shape'] = shape
    dom = 'cw_render_pieces_kernel'
    r = kern.get(dom, {})
    if 'WRITE_SIZE' in r and 'FETCH_SIZE' in r:
        w, f = r['WRITE_SIZE']['median_bytes'], r['FETCH_SIZE']['median_bytes']
        res['hbm_bytes_per_launch'] = w + 2.0 * f
        res['hbm_bytes_per_launch_note'] = dom + ', median launch: WRITE_SIZE + 2 x FETCH_SIZE (gfx950 read correction)'
        if shape:
            S = shape['size']
            frame = 27 * S * (S + 1) if shape['raster'] == 'alt' else 48 * S * S
            # a large batch is swept in several launches of whole multiples of 4 096 envs (cw_kernels.hip: cw_piece_chunks, 896 rounds of 3 KB per wave)
            n, cap = shape['envs_per_gpu'], 896 * 1024 * 3072
            chunks = max(1, -(-n * frame // cap))
            per = -(-n // chunks)
            if chunks > 1:
                per = -(-per // 4096) * 4096
                chunks = -(-n // per)
            res['launches_per_sweep'] = chunks
            res['envs_per_launch'] = min(per, n)
            res['algorithmic_bytes_per_launch'] = min(per, n) * float(S * S + frame)      # (the median launch is a full chunk)
            res['traffic_over_algorithmic'] = res['hbm_bytes_per_launch'] / res['algorithmic_bytes_per_launch']
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    main(sys.argv[1])
