"""CPU tier: what THIS toolchain makes of the HIP kernels -- registers, spills, LDS, and where the full-frame sweep's batch loop
lies in the gfx950 code object (its placement modulo 32 bytes moves the kernel's launch time by up to 17 %, profiles/history/r02_pace.txt N-P,
which is why the one-launch step is built at all eight placements and the engine measures which one to run).  The table is committed
(profiles/r03_isa_resources.txt); a compiler or source change that moves any of it fails here and is visible in review:
regenerate with `python tools/isa_report.py --write` and re-run the placement table (tools/microbench/specs/r03_placement.spec)."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _report():
    spec = importlib.util.spec_from_file_location('isa_report', os.path.join(ROOT, 'tools', 'isa_report.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod, mod.report()


def test_resource_table_and_sweep_loop_placements_match_the_committed_record():
    mod, text = _report()
    rows = {l.split()[0]: l.split() for l in text.splitlines() if l and not l.startswith('#') and not l.startswith('kernel')}
    # the one-launch step exists at every placement of its batch loop modulo 32 bytes, one s_nop (4 bytes) apart
    for k in range(8):
        assert rows['cw_render_step_kernel<%d>' % k][-1] == str(4 * k), rows['cw_render_step_kernel<%d>' % k]
    # ... and they are the same kernel otherwise (registers, spills, LDS)
    assert len({tuple(rows['cw_render_step_kernel<%d>' % k][1:8]) for k in range(8)}) == 1
    # no kernel spills VGPRs or uses scratch memory
    for name, f in rows.items():
        assert f[4] == '0' and f[5] == '0', (name, f)
    with open(mod.RECORD) as fh:
        recorded = fh.read()
    assert text == recorded, ('the build differs from profiles/r03_isa_resources.txt (toolchain or kernel source changed):\n' + text +
                              '\nregenerate with `python tools/isa_report.py --write`, then re-measure the placements '
                              '(tools/microbench/specs/r03_placement.spec) and the perf floor (pytest -m gpu -k perf_floor)')
