#!/usr/bin/env python3
"""Per-kernel resource table of the HIP engine as built by THIS toolchain.

    python tools/isa_report.py            print the table
    python tools/isa_report.py --write    ... and rewrite profiles/r06_isa_resources.txt (tests/test_isa.py prints a warning when a build differs)

Source: hipcc -Rpass-analysis=kernel-resource-usage (registers, spills, scratch, occupancy, LDS)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'gym_craftingworld_amd', 'csrc')
RECORD = os.path.join(ROOT, 'profiles', 'r06_isa_resources.txt')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '--cuda-device-only', '-x', 'hip']


def demangle(names):
    """_Z23cw_render_pieces_kernelILi0ELi2EEv8CwParamsPhiiiii -> cw_render_pieces_kernel<0,2> (Itanium names of plain / int- and bool-templated functions)"""
    out = []
    for n in names:
        m = re.match(r'_Z(\d+)', n)
        if not m:
            out.append(n)
            continue
        k = int(m.group(1))
        base, rest = n[m.end():m.end() + k], n[m.end() + k:]
        t = re.match(r'I((?:L[ib]\d+E)+)E', rest)
        out.append(base + ('<%s>' % ','.join(re.findall(r'L[ib](\d+)E', t.group(1))) if t else ''))
    return out


def resources(tmp):
    r = subprocess.run([HIPCC] + FLAGS + ['-S', '-o', os.path.join(tmp, 'k.s'), os.path.join(CSRC, 'cw_kernels.hip'),
                        '-Rpass-analysis=kernel-resource-usage'], capture_output=True, text=True, check=True)
    rows, cur = [], None
    for line in r.stderr.split('\n'):
        m = re.search(r'remark: +Function Name: (\S+)', line)
        if m:
            cur = {'name': m.group(1)}
            rows.append(cur)
            continue
        m = re.search(r'remark: +([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+) \[-Rpass', line)
        if m and cur is not None:
            cur[m.group(1).strip()] = m.group(2)
    return rows


def report():
    with tempfile.TemporaryDirectory() as tmp:
        rows = resources(tmp)
    names = demangle([r['name'] for r in rows])
    lines = ['# built by tools/isa_report.py (hipcc -O3 --offload-arch=gfx950); tests/test_isa.py warns when a build differs from this table',
             '%-44s %5s %5s %11s %11s %8s %5s %6s' % ('kernel', 'VGPR', 'SGPR', 'SGPR spills', 'VGPR spills', 'scratch', 'occ', 'LDS')]
    for r, n in sorted(zip(rows, names), key=lambda t: t[1]):
        lines.append('%-44s %5s %5s %11s %11s %8s %5s %6s' % (n, r.get('VGPRs', '?'), r.get('TotalSGPRs', '?'), r.get('SGPRs Spill', '?'),
                                                              r.get('VGPRs Spill', '?'), r.get('ScratchSize', '?'), r.get('Occupancy', '?'),
                                                              r.get('LDS Size', '?')))
    return '\n'.join(lines) + '\n'


if __name__ == '__main__':
    text = report()
    sys.stdout.write(text)
    if '--write' in sys.argv:
        with open(RECORD, 'w') as f:
            f.write(text)
