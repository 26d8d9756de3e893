"""CPU tier: what THIS toolchain makes of the HIP kernels -- registers, spills, LDS (tools/isa_report.py; the table of the committed build is
profiles/r06_isa_resources.txt)."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _report():
    spec = importlib.util.spec_from_file_location('isa_report', os.path.join(ROOT, 'tools', 'isa_report.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod, mod.report()


def test_no_kernel_spills_vector_registers_or_uses_scratch():
    """Hard assertions are structural only (a hipcc bump must not fail the CPU tier): no kernel spills VGPRs or touches scratch memory.  The
    committed table (profiles/r06_isa_resources.txt) is compared for information: a difference is printed, not failed."""
    import warnings
    mod, text = _report()
    rows = {l.split()[0]: l.split() for l in text.splitlines() if l and not l.startswith('#') and not l.startswith('kernel')}
    assert 'cw_render_pieces_kernel<0,2>' in rows and 'cw_render_pieces_kernel<1,16>' in rows and 'cw_step_kernel' in rows and 'cw_refill_kernel' in rows
    for name, f in rows.items():
        assert f[4] == '0' and f[5] == '0', (name, f)
    if os.path.exists(mod.RECORD):
        with open(mod.RECORD) as fh:
            if fh.read() != text:
                warnings.warn('the build differs from %s (toolchain or kernel source changed): regenerate with `python tools/isa_report.py '
                              '--write` and re-run the perf floors (pytest -m gpu -k perf_floor)\n%s' % (os.path.relpath(mod.RECORD, ROOT), text))
