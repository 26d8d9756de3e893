"""Which build of the engine's HIP-free host logic (csrc/cw_host.cpp) the CPU tests call: the product library, or -- with CW_HOST_LIB set, as
tests/test_sanitizers.py does in a subprocess under LD_PRELOAD=libasan -- the ASAN/UBSAN build of cw_host.cpp alone (make -C
gym_craftingworld_amd/csrc host_asan).  The sanitizer run must not import torch (a preloaded libasan and the ROCm runtime do not mix), so the
binding module is loaded by path there, without the package around it."""
import ctypes as C
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_cache = {}


def host_lib():
    """-> (binding module, library with every cwh_* symbol bound)"""
    path = os.environ.get('CW_HOST_LIB', '')
    if path not in _cache:
        if path:
            spec = importlib.util.spec_from_file_location('cw_lib_binding', os.path.join(ROOT, 'gym_craftingworld_amd', '_lib.py'))
            L = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(L)
            _cache[path] = (L, L.bind_host_helpers(C.CDLL(path)))
        else:
            import __graft_entry__ as g
            g.build()
            from gym_craftingworld_amd import _lib as L
            _cache[path] = (L, L.load())
    return _cache[path]
