"""ctypes binding of libcraftingworld.so (C ABI: include/craftingworld.h).

The library is the product: there is no Python/CPU fallback.  If the shared object is missing
this module raises at import of the engine (build it with `python -c "import __graft_entry__ as g;
g.build()"` or `make -C gym_craftingworld_amd/csrc`).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libcraftingworld.so')


def lib_path():
    """Which build to load: CW_LIB_PATH (any other build, A/B runs) or the product."""
    return os.environ.get('CW_LIB_PATH') or LIB_PATH

CW_ABI_VERSION = 5
CW_MT_N = 624
CW_MAX_TASKS = 16
CW_MAX_MENUS = 256

CW_OK, CW_ERR_INVALID, CW_ERR_HIP, CW_ERR_STATE = 0, -1, -2, -3
CW_OBS_STATE, CW_OBS_PIXELS_FULL, CW_OBS_PIXELS_DIRTY = 0, 1, 2
CW_ACT_I32, CW_ACT_I64, CW_ACT_U8 = 0, 1, 2
CW_RASTER_RAY, CW_RASTER_ALT = 0, 1


class cw_task_menu(C.Structure):
    _fields_ = [('n_selected', C.c_int32), ('number_of_tasks', C.c_int32), ('stacking', C.c_int32),
                ('reward_subset', C.c_int32), ('selected_bits', C.c_int32 * CW_MAX_TASKS)]


class cw_config(C.Structure):
    _fields_ = [('abi_version', C.c_int32), ('num_envs', C.c_int32), ('size', C.c_int32),
                ('max_steps', C.c_int32), ('n_task_list', C.c_int32), ('fixed_init_state', C.c_int32),
                ('obs_mode', C.c_int32), ('auto_reset', C.c_int32), ('keep_terminal_obs', C.c_int32),
                ('raster', C.c_int32), ('host_outputs', C.c_int32), ('n_menus', C.c_int32),
                ('menus', C.POINTER(cw_task_menu)), ('env_menu', C.POINTER(C.c_uint8))]


class cw_buffer_table(C.Structure):
    _fields_ = [('obs', C.c_void_p), ('desired_goal', C.c_void_p), ('init_obs', C.c_void_p), ('terminal_obs', C.c_void_p),
                ('reward', C.c_void_p), ('done', C.c_void_p), ('achieved', C.c_void_p),
                ('desired', C.c_void_p), ('episode_length', C.c_void_p), ('episode_return', C.c_void_p), ('hdr', C.c_void_p),
                ('slot_pos', C.c_void_p), ('counters', C.c_void_p), ('frame_bytes', C.c_size_t),
                ('host_actions', C.c_void_p), ('host_onehot', C.c_void_p)]


class cw_state_view(C.Structure):
    _fields_ = [('grid', C.c_void_p), ('init_grid', C.c_void_p), ('goal_grid', C.c_void_p),
                ('agent_rc', C.c_void_p), ('init_agent_rc', C.c_void_p), ('goal_agent_rc', C.c_void_p),
                ('hold', C.c_void_p), ('achieved', C.c_void_p), ('desired', C.c_void_p),
                ('step_num', C.c_void_p), ('ep_no', C.c_void_p)]


class cw_profile(C.Structure):
    _fields_ = [('steps', C.c_int32), ('ms_step_kernel', C.c_float), ('ms_reset_kernel', C.c_float),
                ('ms_render_kernel', C.c_float), ('ms_render_kernel_max', C.c_float), ('ms_render_kernel_min', C.c_float),
                ('ms_render_kernel_median', C.c_float)]


class cw_tuner_state(C.Structure):
    _fields_ = [('period16', C.c_int32), ('period16_head', C.c_int32), ('period16_busy', C.c_int32), ('lookahead', C.c_int32), ('resident', C.c_int32),
                ('guard_slowdowns', C.c_int32)]


# every symbol include/craftingworld.h declares: name -> (restype, argtypes)
_VP = C.c_void_p
ABI = {
    'cw_create': (C.c_int, [C.POINTER(cw_config), C.c_int, C.POINTER(_VP)]),
    'cw_destroy': (C.c_int, [_VP]),
    'cw_seed_mt': (C.c_int, [_VP, _VP, _VP]),
    'cw_seed_int': (C.c_int, [_VP, _VP]),
    'cw_get_mt': (C.c_int, [_VP, _VP, _VP]),
    'cw_generate_fixed_states': (C.c_int, [_VP, _VP]),
    'cw_get_fixed_states': (C.c_int, [_VP, _VP]),
    'cw_reset': (C.c_int, [_VP, _VP]),
    'cw_step': (C.c_int, [_VP, _VP, C.c_int, _VP]),
    'cw_step_many': (C.c_int, [_VP, _VP, C.c_int, C.c_int32, _VP]),
    'cw_rollout': (C.c_int, [_VP, _VP, C.c_int32, _VP, _VP, _VP]),
    'cw_render': (C.c_int, [_VP, _VP, _VP]),
    'cw_render_onehot': (C.c_int, [_VP, _VP, C.c_int32, _VP, _VP]),
    'cw_export_grid': (C.c_int, [_VP, _VP, _VP]),
    'cw_export_onehot': (C.c_int, [_VP, _VP, _VP]),
    'cw_export_onehot_of': (C.c_int, [_VP, C.c_int, _VP, _VP]),
    'cw_get_state': (C.c_int, [_VP, C.POINTER(cw_state_view)]),
    'cw_set_state': (C.c_int, [_VP, C.POINTER(cw_state_view)]),
    'cw_checkpoint_bytes': (C.c_size_t, [_VP]),
    'cw_checkpoint_save': (C.c_int, [_VP, _VP, C.c_size_t]),
    'cw_checkpoint_load': (C.c_int, [_VP, _VP, C.c_size_t]),
    'cw_profile_begin': (C.c_int, [_VP, C.c_int]),
    'cw_profile_end': (C.c_int, [_VP, C.POINTER(cw_profile)]),
    'cw_tuner': (C.c_int, [_VP, C.POINTER(cw_tuner_state)]),
    'cw_step_resident': (C.c_int, [_VP, C.c_int32, C.c_int32]),
    'cw_resident_stop': (C.c_int, [_VP]),
    'cw_render_kernel_name': (C.c_char_p, [_VP]),
    'cw_buffers': (C.c_int, [_VP, C.POINTER(cw_buffer_table)]),
    'cw_synchronize': (C.c_int, [_VP, _VP]),
    'cw_num_envs': (C.c_int, [_VP]),
    'cw_abi_version': (C.c_int, []),
    'cw_last_error': (C.c_char_p, []),
}
class cwh_guard(C.Structure):
    """the sweep clock's guard as a pure state machine (csrc/cw_host.h)"""
    _fields_ = [('rate', C.c_double), ('rate_top', C.c_double), ('ms_sum', C.c_double), ('prev_mean', C.c_double), ('ref_ms', C.c_double),
                ('ref_prev', C.c_double), ('ms_n', C.c_int32), ('late', C.c_int32), ('good', C.c_int32), ('slowdowns', C.c_int32),
                ('probes', C.c_int32), ('probe_need', C.c_int32), ('recover_need', C.c_int32), ('probing', C.c_int32), ('recovering', C.c_int32)]


CWH_GUARD_NONE, CWH_GUARD_SLOWDOWN, CWH_GUARD_TRIAL_UP, CWH_GUARD_TRIAL_KEPT, CWH_GUARD_TRIAL_UNDONE = range(5)
CWH_CKPT_SECTIONS = 22

# the engine's HIP-free host logic (csrc/cw_host.h: MT19937 state conversion, DLPack, dense views, checkpoint sizes, the guard's decisions), exported for
# the tests of the host logic -- the same table binds libcw_host_asan.so, the ASAN/UBSAN build of cw_host.cpp alone (bind_host_helpers)
HOST_HELPERS = {
    'cwh_mt_from_numpy': (C.c_int, [_VP, C.c_int]),
    'cwh_mt_to_numpy': (None, [_VP, C.c_int, _VP]),
    'cwh_mt_init_genrand': (None, [_VP, C.c_uint32]),
    'cwh_mt_untwist': (None, [_VP]),
    'cwh_mt_rewind': (None, [_VP, C.POINTER(C.c_int32), C.c_uint32]),
    'cwh_dlpack_make': (_VP, [_VP, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int64)]),
    'cwh_slots_to_grid': (None, [_VP, C.c_uint32, C.c_int, _VP]),
    'cwh_ckpt_section_bytes': (C.c_int, [C.c_int64, C.c_int32, C.c_int32, C.POINTER(C.c_size_t), C.POINTER(C.c_uint64)]),
    'cwh_guard_init': (None, [C.POINTER(cwh_guard), C.c_double]),
    'cwh_guard_step': (C.c_int, [C.POINTER(cwh_guard), C.c_double, C.c_double]),
    'cwh_sweep_periods': (None, [C.c_double, C.c_int32, C.c_double, C.c_double, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    'cwh_guard_scheduled_ms': (C.c_double, [C.c_double, C.c_int32, C.c_int32, C.c_double]),
    'cwh_la_adapt': (C.c_int32, [C.c_int32, C.c_int32, C.c_uint64, C.POINTER(C.c_int32)]),
}


def bind_host_helpers(lib):
    """argtypes / restype of every cwh_* symbol on `lib` (the product, or the sanitizer build of cw_host.cpp)"""
    for name, (res, args) in HOST_HELPERS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib

_libs = {}      # path -> loaded library (a process may hold the product and another build side by side)
_lib = None     # the one loaded (or asked for) last: check() takes its error text


class CraftingWorldError(RuntimeError):
    pass


def load():
    """Load the engine library chosen by lib_path() (after torch, so both share one HIP runtime); cached per path."""
    global _lib
    path = lib_path()
    if path in _libs:
        _lib = _libs[path]
        return _lib
    if not os.path.exists(path):
        raise CraftingWorldError(
            'HIP extension %s is missing; build it (python -c "import __graft_entry__ as g; g.build()"). '
            'There is no CPU fallback.' % path)
    try:
        import torch  # noqa: F401  -- loads torch's libamdhip64 first so the SONAME resolves to it
    except ImportError:
        pass
    lib = C.CDLL(path)
    for table in (ABI, HOST_HELPERS):
        for name, (res, args) in table.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
    if lib.cw_abi_version() != CW_ABI_VERSION:
        raise CraftingWorldError('%s ABI %d != binding %d' % (os.path.basename(path), lib.cw_abi_version(), CW_ABI_VERSION))
    _libs[path] = _lib = lib
    return lib


def check(rc, what, lib=None):
    """rc of a library call -> None, ValueError (CW_ERR_INVALID) or CraftingWorldError; `lib`: the library the call went to (a process may hold
    several builds: the error text is that library's)"""
    if rc == CW_OK:
        return
    msg = ((lib or _lib or load()).cw_last_error() or b'').decode(errors='replace')
    if rc == CW_ERR_INVALID:
        raise ValueError('%s: %s' % (what, msg))
    raise CraftingWorldError('%s failed (%d): %s' % (what, rc, msg))
