"""ctypes wrapper of the C oracle (oracle/cw_oracle.c) -- TEST INFRASTRUCTURE ONLY.

`OracleEnv` mirrors the reference's CraftingWorldEnvRay surface closely enough that the parity
tests read like reference tests (reset() -> dict of four images, step(a) -> (obs, reward, done,
info)); `OracleBatch` is N independent OracleEnvs with gym.vector-style auto-reset, the CPU
counterpart of the HIP engine's batch semantics.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

TASK_LIST = ['MakeBread', 'EatBread', 'BuildHouse', 'ChopTree', 'ChopRock', 'GoToHouse', 'MoveAxe',
             'MoveHammer', 'MoveSticks']  # ray.py:40-41

MT_N = 624
MAX_TASKS = 16


class _Config(C.Structure):
    _fields_ = [('size', C.c_int32), ('max_steps', C.c_int32), ('reward_subset', C.c_int32),
                ('stacking', C.c_int32), ('n_task_list', C.c_int32), ('n_selected', C.c_int32),
                ('number_of_tasks', C.c_int32), ('fixed_init_state', C.c_int32),
                ('selected_bits', C.c_int32 * MAX_TASKS), ('alt_obs', C.c_int32)]


class _View(C.Structure):
    _fields_ = [('grid', C.POINTER(C.c_uint8)), ('init_grid', C.POINTER(C.c_uint8)),
                ('goal_grid', C.POINTER(C.c_uint8)), ('obs', C.POINTER(C.c_uint8)),
                ('desired_img', C.POINTER(C.c_uint8)), ('init_img', C.POINTER(C.c_uint8)),
                ('agent_r', C.c_int32), ('agent_c', C.c_int32), ('hold', C.c_int32),
                ('goal_agent_r', C.c_int32), ('goal_agent_c', C.c_int32),
                ('init_agent_r', C.c_int32), ('init_agent_c', C.c_int32),
                ('achieved', C.c_uint32), ('desired', C.c_uint32),
                ('step_num', C.c_int32), ('ep_no', C.c_int32)]


def build_oracle(force=False):
    """Compile oracle/libcw_oracle.so with gcc if missing or stale (no GPU, no reference needed)."""
    if os.environ.get('CW_ORACLE_SO'):            # e.g. the ASAN/UBSAN build from `make -C oracle asan`
        return os.environ['CW_ORACLE_SO']
    so = os.path.join(_HERE, 'libcw_oracle.so')
    srcs = [os.path.join(_HERE, f) for f in ('cw_oracle.c', 'cw_oracle.h')]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(['make', '-C', _HERE, 'libcw_oracle.so'], stdout=subprocess.DEVNULL)
    return so


def _lib():
    global _LIB
    if _LIB is None:
        lib = C.CDLL(build_oracle())
        vp, u32p, i32p, u8p = C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_int32), C.POINTER(C.c_uint8)
        lib.cwo_new.restype = vp
        lib.cwo_new.argtypes = [C.POINTER(_Config)]
        lib.cwo_free.argtypes = [vp]
        lib.cwo_set_rng.argtypes = [vp, u32p, C.c_int32]
        lib.cwo_get_rng.argtypes = [vp, u32p, i32p]
        lib.cwo_seed_int.argtypes = [vp, C.c_uint32]
        lib.cwo_rng_u32.restype = C.c_uint32
        lib.cwo_rng_u32.argtypes = [vp]
        lib.cwo_rng_randint.restype = C.c_uint32
        lib.cwo_rng_randint.argtypes = [vp, C.c_uint32]
        lib.cwo_rng_shuffle.argtypes = [vp, i32p, C.c_int32]
        lib.cwo_generate_fixed_states.argtypes = [vp]
        lib.cwo_get_fixed_states.argtypes = [vp, C.POINTER(C.c_uint16)]
        lib.cwo_reset.argtypes = [vp]
        lib.cwo_step.restype = C.c_int
        lib.cwo_step.argtypes = [vp, C.c_int32, i32p, i32p]
        lib.cwo_get_view.argtypes = [vp, C.POINTER(_View)]
        lib.cwo_set_state.argtypes = [vp, u8p, u8p, C.c_int32, C.c_int32, C.c_int32, C.c_uint32,
                                      C.c_uint32, C.c_int32]
        lib.cwo_render.argtypes = [C.c_int32, u8p, C.c_int32, C.c_int32, C.c_int32, u8p]
        lib.cwo_batch_rollout_full.restype = C.c_int64
        lib.cwo_batch_rollout_full.argtypes = [C.POINTER(vp), C.c_int32, C.POINTER(C.c_int8), C.c_int32, C.c_int32]
        lib.cwo_batch_rollout.restype = C.c_int64
        lib.cwo_batch_rollout.argtypes = [C.POINTER(vp), C.c_int32, C.POINTER(C.c_int8), C.c_int32,
                                          C.c_int32, i32p, u8p]
        _LIB = lib
    return _LIB


def _bits_to_vec(bits, n):
    return np.array([[(bits >> i) & 1 for i in range(n)]], dtype=np.int64)


def make_config(size=(21, 21), fixed_init_state=0, max_steps=300, task_list=TASK_LIST,
                selected_tasks=TASK_LIST, number_of_tasks=None, stacking=True, reward_style=None, alt_obs=False):
    """Reference ctor kwargs (ray.py:59-60; alt_obs selects CraftingWorldEnvAltObs's rasteriser) -> cwo_config."""
    w, h = size
    if w != h:
        raise ValueError('non-square grids are a reference defect (SURVEY.md §8a) and are rejected')
    cfg = _Config()
    cfg.size = w
    cfg.max_steps = max_steps
    cfg.reward_subset = 0 if reward_style is None else 1
    cfg.stacking = 1 if stacking is True else 0          # `stacking is True`, ray.py:169
    cfg.n_task_list = len(task_list)
    cfg.n_selected = len(selected_tasks)
    n = number_of_tasks if number_of_tasks is not None else len(selected_tasks)
    cfg.number_of_tasks = min(n, len(selected_tasks))    # ray.py:79-81
    cfg.fixed_init_state = fixed_init_state
    cfg.alt_obs = 1 if alt_obs else 0
    for i, t in enumerate(selected_tasks):
        cfg.selected_bits[i] = list(task_list).index(t)  # ray.py:174
    return cfg


class OracleEnv:
    """Single env, reference-shaped API, backed by the C oracle."""

    def __init__(self, rng_state=None, **kwargs):
        self._lib = _lib()
        self.cfg = make_config(**kwargs)
        self._h = self._lib.cwo_new(C.byref(self.cfg))
        if not self._h:
            raise ValueError('cwo_new rejected the config')
        self.size = self.cfg.size
        self.img_shape = ((3 * self.cfg.size + 3, 3 * self.cfg.size, 3) if self.cfg.alt_obs else
                          (4 * self.cfg.size, 4 * self.cfg.size, 3))
        self.MAX_STEPS = self.cfg.max_steps
        self.n_task_list = self.cfg.n_task_list
        if rng_state is not None:
            self.set_rng(*rng_state)
        if self.cfg.fixed_init_state:
            self._lib.cwo_generate_fixed_states(self._h)

    def __del__(self):
        if getattr(self, '_h', None):
            self._lib.cwo_free(self._h)
            self._h = None

    # -- RNG (numpy RandomState key/pos) --
    def set_rng(self, key, pos):
        key = np.ascontiguousarray(key, dtype=np.uint32)
        assert key.shape == (MT_N,)
        self._lib.cwo_set_rng(self._h, key.ctypes.data_as(C.POINTER(C.c_uint32)), int(pos))

    def get_rng(self):
        key = np.empty(MT_N, dtype=np.uint32)
        pos = C.c_int32()
        self._lib.cwo_get_rng(self._h, key.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(pos))
        return key, pos.value

    def seed_int(self, seed):
        self._lib.cwo_seed_int(self._h, seed)

    def rng_u32(self):
        return self._lib.cwo_rng_u32(self._h)

    def rng_randint(self, n):
        return self._lib.cwo_rng_randint(self._h, n)

    def rng_shuffle(self, n):
        x = np.arange(n, dtype=np.int32)
        self._lib.cwo_rng_shuffle(self._h, x.ctypes.data_as(C.POINTER(C.c_int32)), n)
        return x

    def fixed_states(self):
        """fixed_state_list (ray.py:116-118) as cell indices: uint16 [K, 9] = objects 0..7 then the agent"""
        out = np.zeros((max(self.cfg.fixed_init_state, 1), 9), dtype=np.uint16)
        self._lib.cwo_get_fixed_states(self._h, out.ctypes.data_as(C.POINTER(C.c_uint16)))
        return out

    # -- env --
    def view(self):
        v = _View()
        self._lib.cwo_get_view(self._h, C.byref(v))
        return v

    def _arr(self, ptr, shape):
        return np.ctypeslib.as_array(ptr, shape=shape)

    def state(self):
        """Copies of everything the parity tests compare."""
        v = self.view()
        s, ish = self.size, self.img_shape
        return dict(grid=self._arr(v.grid, (s, s)).copy(), init_grid=self._arr(v.init_grid, (s, s)).copy(),
                    goal_grid=self._arr(v.goal_grid, (s, s)).copy(),
                    agent=(v.agent_r, v.agent_c), hold=v.hold, goal_agent=(v.goal_agent_r, v.goal_agent_c),
                    init_agent=(v.init_agent_r, v.init_agent_c),
                    achieved=v.achieved, desired=v.desired, step_num=v.step_num, ep_no=v.ep_no,
                    obs=self._arr(v.obs, ish).copy(),
                    desired_img=self._arr(v.desired_img, ish).copy(),
                    init_img=self._arr(v.init_img, ish).copy())

    def _obs_dict(self, v=None):
        v = v or self.view()
        ish = self.img_shape
        o = self._arr(v.obs, ish)
        return {'observation': o, 'desired_goal': self._arr(v.desired_img, ish),
                'achieved_goal': o, 'init_observation': self._arr(v.init_img, ish)}

    def reset(self):
        self._lib.cwo_reset(self._h)
        return self._obs_dict()

    def step(self, action):
        r, d = C.c_int32(), C.c_int32()
        if self._lib.cwo_step(self._h, int(action), C.byref(r), C.byref(d)) != 0:
            raise IndexError('action out of range')      # ACTIONS[action], ray.py:308
        v = self.view()
        info = {'task_success': _bits_to_vec(v.achieved, self.n_task_list),
                'desired_goal': _bits_to_vec(v.desired, self.n_task_list),
                'achieved_goal': _bits_to_vec(v.achieved, self.n_task_list)}
        return self._obs_dict(v), r.value, bool(d.value), info

    def set_state(self, grid, init_grid, agent, hold, achieved, desired, step_num):
        g = np.ascontiguousarray(grid, dtype=np.uint8)
        ig = np.ascontiguousarray(init_grid, dtype=np.uint8)
        u8p = C.POINTER(C.c_uint8)
        self._lib.cwo_set_state(self._h, g.ctypes.data_as(u8p), ig.ctypes.data_as(u8p), int(agent[0]),
                                int(agent[1]), int(hold), int(achieved), int(desired), int(step_num))


def render(size, grid, agent, hold):
    """ray.py:442-520 full-frame render of (grid codes, agent, hold) -> (4s,4s,3) uint8."""
    g = np.ascontiguousarray(grid, dtype=np.uint8)
    out = np.empty((size * 4, size * 4, 3), dtype=np.uint8)
    u8p = C.POINTER(C.c_uint8)
    _lib().cwo_render(size, g.ctypes.data_as(u8p), int(agent[0]), int(agent[1]), int(hold),
                      out.ctypes.data_as(u8p))
    return out


class OracleBatch:
    """N independent oracle envs with auto-reset (CPU counterpart of the HIP engine's batch)."""

    def __init__(self, num_envs, rng_states=None, per_env_kwargs=None, **kwargs):
        self.num_envs = num_envs
        self.envs = []
        for i in range(num_envs):
            kw = dict(kwargs)
            if per_env_kwargs is not None:
                kw.update(per_env_kwargs[i])
            self.envs.append(OracleEnv(rng_state=None if rng_states is None else rng_states[i], **kw))
        self._handles = (C.c_void_p * num_envs)(*[e._h for e in self.envs])

    def reset(self):
        for e in self.envs:
            e.reset()

    def step(self, actions):
        """-> (reward[N] int32, done[N] bool); done envs are reset (their obs is the new episode's)."""
        rew = np.empty(self.num_envs, dtype=np.int32)
        done = np.zeros(self.num_envs, dtype=bool)
        for i, e in enumerate(self.envs):
            _, rew[i], done[i], _ = e.step(int(actions[i]))
            if done[i]:
                e.reset()
        return rew, done

    def rollout(self, actions, nthreads=1, record=False):
        """actions int8 [T,N]; returns env-steps executed (and reward/done arrays if record)."""
        a = np.ascontiguousarray(actions, dtype=np.int8)
        T, n = a.shape
        assert n == self.num_envs
        rew = np.empty((T, n), dtype=np.int32) if record else None
        don = np.empty((T, n), dtype=np.uint8) if record else None
        total = _lib().cwo_batch_rollout(
            self._handles, n, a.ctypes.data_as(C.POINTER(C.c_int8)), T, nthreads,
            rew.ctypes.data_as(C.POINTER(C.c_int32)) if record else None,
            don.ctypes.data_as(C.POINTER(C.c_uint8)) if record else None)
        return (total, rew, don) if record else total

    def rollout_full_frame(self, actions, nthreads=1):
        """Like rollout(record=False), but with a full render() of every env's frame after every step."""
        a = np.ascontiguousarray(actions, dtype=np.int8)
        T, n = a.shape
        assert n == self.num_envs
        return _lib().cwo_batch_rollout_full(self._handles, n, a.ctypes.data_as(C.POINTER(C.c_int8)), T, nthreads)

    def states(self):
        return [e.state() for e in self.envs]
