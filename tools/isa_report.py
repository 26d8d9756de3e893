#!/usr/bin/env python3
"""Per-kernel resource table of the HIP engine as built by THIS toolchain, and where the sweep's batch loop lies in the code object.

    python tools/isa_report.py            print the table
    python tools/isa_report.py --write    ... and rewrite profiles/r03_isa_resources.txt (tests/test_isa.py compares against it, so a
                                          compiler or source change that moves registers, spills or the loop shows up in review)

Sources: hipcc -Rpass-analysis=kernel-resource-usage (registers, spills, scratch, occupancy, LDS) and the symbol table of the gfx950 code
object (render_groups marks its batch loop with a local symbol cw_sweep_head_<n>; the one-launch step is built at the eight placements
of that loop modulo 32 bytes, cw_render_step_kernel<0..7>, and cw_step measures which one to run -- DESIGN.md 4.3)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'gym_craftingworld_amd', 'csrc')
RECORD = os.path.join(ROOT, 'profiles', 'r03_isa_resources.txt')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
LLVM = '/opt/rocm/lib/llvm/bin'
FLAGS = ['-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '--cuda-device-only', '-x', 'hip']


def demangle(names):
    """_Z21cw_render_step_kernelILi3EEv8CwParamsii -> cw_render_step_kernel<3> (Itanium names of plain / int-templated functions)"""
    out = []
    for n in names:
        m = re.match(r'_Z(\d+)', n)
        if not m:
            out.append(n)
            continue
        k = int(m.group(1))
        base, rest = n[m.end():m.end() + k], n[m.end() + k:]
        t = re.match(r'ILi(\d+)E', rest)
        out.append(base + ('<%s>' % t.group(1) if t else ''))
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


def loop_heads(tmp):
    obj, co = os.path.join(tmp, 'k.o'), os.path.join(tmp, 'k.co')
    subprocess.run([HIPCC] + FLAGS + ['-c', '-o', obj, os.path.join(CSRC, 'cw_kernels.hip')], capture_output=True, check=True)
    subprocess.run([os.path.join(LLVM, 'clang-offload-bundler'), '--unbundle', '--type=o', '--input=' + obj,
                    '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', '--output=' + co], capture_output=True, check=True)
    syms = subprocess.run([os.path.join(LLVM, 'llvm-objdump'), '--syms', co], capture_output=True, text=True, check=True).stdout
    funcs, heads = [], []
    for line in syms.split('\n'):
        f = line.split()
        if len(f) >= 5 and '.text' in f and f[-1].startswith('cw_sweep_head_'):
            heads.append(int(f[0], 16))
        elif len(f) >= 6 and 'F' in f[1:3] and '.text' in f:
            funcs.append((int(f[0], 16), int(f[4], 16), f[-1]))
    out = {}
    for h in heads:
        for a, size, name in funcs:
            if a <= h < a + size:
                out.setdefault(name, []).append(h)
    return out


def report():
    with tempfile.TemporaryDirectory() as tmp:
        rows = resources(tmp)
        heads = loop_heads(tmp)
    names = demangle([r['name'] for r in rows])
    lines = ['# built by tools/isa_report.py (hipcc -O3 --offload-arch=gfx950); tests/test_isa.py fails when a build differs from this table',
             '# "sweep loop @" = address of render_groups\' batch loop in the code object modulo 32 bytes (its placement: DESIGN.md 4.3)',
             '%-44s %5s %5s %11s %11s %8s %5s %6s  %s' % ('kernel', 'VGPR', 'SGPR', 'SGPR spills', 'VGPR spills', 'scratch', 'occ', 'LDS', 'sweep loop @')]
    for r, n in sorted(zip(rows, names), key=lambda t: t[1]):
        h = ' '.join('%d' % (a % 32) for a in sorted(heads.get(r['name'], [])))
        lines.append('%-44s %5s %5s %11s %11s %8s %5s %6s  %s' % (n, r.get('VGPRs', '?'), r.get('TotalSGPRs', '?'), r.get('SGPRs Spill', '?'),
                                                                  r.get('VGPRs Spill', '?'), r.get('ScratchSize', '?'), r.get('Occupancy', '?'),
                                                                  r.get('LDS Size', '?'), h or '-'))
    return '\n'.join(lines) + '\n'


if __name__ == '__main__':
    text = report()
    sys.stdout.write(text)
    if '--write' in sys.argv:
        with open(RECORD, 'w') as f:
            f.write(text)
