"""CraftingWorldVecEnv -- N CraftingWorld envs stepped by hand-written HIP kernels on one MI355X.

Mirrors the reference's CraftingWorldEnvRay (gym_craftingworld/envs/craftingworld_ray.py,
"ray.py") batched the gym.vector.VectorEnv way: same ctor kwargs (ray.py:59-60), same action
ids (ray.py:130-131), same rewards/done rule (ray.py:361-367), same four-image Dict observation
(ray.py:194-196), auto-reset of finished envs.  All compute happens in libcraftingworld.so
(include/craftingworld.h); this file is host plumbing: config marshalling, torch views of the
engine's device buffers, stream hand-off.  There is no CPU fallback.
"""
import atexit
import ctypes as C
import sys
import weakref

import numpy as np
import torch

from . import _lib as L
from . import seeding
from ._wrap import tensor_view
from .spaces import Box, Dict, Discrete, MultiDiscrete, batch_space

TASK_LIST = ['MakeBread', 'EatBread', 'BuildHouse', 'ChopTree', 'ChopRock', 'GoToHouse', 'MoveAxe',
             'MoveHammer', 'MoveSticks']                       # ray.py:40-41
OBJECTS = ['sticks', 'axe', 'hammer', 'rock', 'tree', 'bread', 'house', 'wheat']   # ray.py:21
PICKUPABLE = ['sticks', 'axe', 'hammer']                       # ray.py:20
ACTION_NAMES = ['up', 'right', 'down', 'left', 'pickup', 'drop']   # ray.py:130-131
UP, RIGHT, DOWN, LEFT, PICKUP, DROP = range(6)                  # action ids (ray.py:15-18 names the four moves; 4 / 5 are pickup / drop, ray.py:130-131)
# the palette (ray.py:26-31), as data for callers that decode frames: an object's 4x4 tile, the same with index 0 = the empty floor, and what the
# reference SUBTRACTS from the agent's white centre for a held item (255 - its colour).  The kernels carry their own copy (csrc/cw_kernels.hip).
COLORS = [(110, 69, 39), (255, 105, 180), (100, 100, 200), (100, 100, 100), (0, 128, 0), (205, 133, 63), (197, 91, 97), (240, 230, 140)]
COLORS_N = [(0, 0, 0)] + COLORS
COLORS_H = [tuple(255 - c for c in COLORS[k]) for k in range(3)]
STATE_W = STATE_H = 21                                         # ray.py:43-44: the default grid
MAX_STEPS = 300                                                # ray.py:46

_OBS_MODES = {'state': L.CW_OBS_STATE, 'pixels': L.CW_OBS_PIXELS_FULL, 'pixels_dirty': L.CW_OBS_PIXELS_DIRTY}
_ACT_DTYPES = {torch.int32: L.CW_ACT_I32, torch.int64: L.CW_ACT_I64, torch.uint8: L.CW_ACT_U8}


def _menu_struct(task_list, selected_tasks, number_of_tasks, stacking, reward_style):
    m = L.cw_task_menu()
    selected_tasks = list(selected_tasks)
    if not 1 <= len(selected_tasks) <= L.CW_MAX_TASKS:
        raise ValueError('selected_tasks must hold 1..%d tasks' % L.CW_MAX_TASKS)
    m.n_selected = len(selected_tasks)
    n = number_of_tasks if number_of_tasks is not None else len(selected_tasks)      # ray.py:79
    m.number_of_tasks = min(int(n), len(selected_tasks))                             # ray.py:80-81
    if m.number_of_tasks < 1:
        raise ValueError('number_of_tasks must be >= 1')
    m.stacking = 1 if stacking is True else 0                                        # `is True`, ray.py:169
    m.reward_subset = 0 if reward_style is None else 1                               # ray.py:71-74
    tl = list(task_list)
    for i, t in enumerate(selected_tasks):
        m.selected_bits[i] = tl.index(t)                                             # ValueError like ray.py:174
    return m


_NP_OF = {torch.uint8: np.uint8, torch.int16: np.int16, torch.int32: np.int32, torch.int64: np.int64}


def _host_view(ptr, shape, dtype):
    """CPU tensor over engine-owned pinned host memory (cw_config.host_outputs); None for a NULL pointer."""
    if not ptr:
        return None
    npdt = np.dtype(_NP_OF[dtype])
    nbytes = int(np.prod(shape)) * npdt.itemsize
    arr = np.frombuffer((C.c_uint8 * nbytes).from_address(int(ptr)), dtype=npdt).reshape(shape)
    return torch.from_numpy(arr)


_LIVE = weakref.WeakSet()


@atexit.register
def _close_all():      # destroy engines before the HIP runtime is torn down at interpreter exit
    for env in list(_LIVE):
        try:
            env.close()
        except Exception:  # noqa: BLE001
            pass


class CraftingWorldVecEnv:
    """Batched drop-in for CraftingWorldEnvRay; see module docstring.

    Extra (non-reference) kwargs: num_envs, obs_mode ('pixels' full-frame render every step |
    'pixels_dirty' persistent frame with <=2 repainted cells per step, the reference's own
    render_edit strategy | 'state' no pixel buffers), device, seed, seed_style, auto_reset,
    task_menus + env_menu (heterogeneous ordered task lists: env i uses task_menus[env_menu[i]],
    each menu a dict with any of selected_tasks / number_of_tasks / stacking / reward_style),
    raster ('ray' = CraftingWorldEnvRay's 4x4 colour tiles, 'alt' = CraftingWorldEnvAltObs's 3x3 CPV tiles),
    keep_terminal_obs (pixel modes: info['terminal_observation'] holds the last frame of every episode
    that ended this step, as gym.vector does, at the cost of one extra frame write per finished env),
    host_outputs (small batches driven from host code -- the single-env gym loop: frames, reward, done and masks
    live in pinned host memory the kernels write directly; they come back as CPU tensors, step()/reset() return
    after one stream sync and there is no copy; actions may be plain ints/arrays).
    """

    metadata = {'render.modes': ['Non']}
    is_vector_env = True          # gym.vector.VectorEnv marker (wrappers key on it)
    viewer = None

    def __init__(self, num_envs, size=(21, 21), fixed_init_state=0, max_steps=300, store_gif=False,
                 render_save_rate=1, task_list=TASK_LIST, selected_tasks=TASK_LIST, number_of_tasks=None,
                 stacking=True, reward_style=None, obs_mode='pixels', device=None, seed=None,
                 seed_style='numpy', auto_reset=True, task_menus=None, env_menu=None,
                 keep_terminal_obs=False, raster='ray', host_outputs=False):
        if store_gif:
            raise NotImplementedError('the GIF episode recorder (ray.py:565-597) records ONE env: use the N=1 classes (env.py, recorder.py) or '
                                      'recorder.EpisodeRecorder on a row of this batch')
        w, h = size
        if w != h:
            raise ValueError('non-square grids raise IndexError in the reference (SURVEY.md §8a); rejected')
        if raster not in ('ray', 'alt'):
            raise ValueError("raster must be 'ray' (4x4 colour tiles) or 'alt' (CraftingWorldEnvAltObs 3x3 CPV tiles)")
        if obs_mode not in _OBS_MODES:
            raise ValueError('obs_mode must be one of %s' % sorted(_OBS_MODES))
        if not torch.cuda.is_available():
            raise L.CraftingWorldError('no MI355X visible to torch: the CraftingWorld engine has no CPU fallback')
        self._lib = L.load()
        dev = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        if dev.type != 'cuda':
            raise ValueError('device must be a cuda (HIP) device')
        self.device = torch.device('cuda', dev.index if dev.index is not None else torch.cuda.current_device())
        self.num_envs = int(num_envs)
        self.STATE_W = self.STATE_H = self.size = int(w)
        self.MAX_STEPS = int(max_steps)
        self.task_list = list(task_list)
        self.selected_tasks = list(selected_tasks)
        self.stacking = stacking
        self.fixed_init_state = int(fixed_init_state)
        self.obs_mode = obs_mode
        self.auto_reset = bool(auto_reset)
        self.seed_style = seed_style

        defaults = dict(selected_tasks=selected_tasks, number_of_tasks=number_of_tasks, stacking=stacking,
                        reward_style=reward_style)
        menus = [dict(defaults)] if task_menus is None else [dict(defaults, **m) for m in task_menus]
        self._menus = (L.cw_task_menu * len(menus))(*[_menu_struct(self.task_list, **m) for m in menus])
        self.number_of_tasks = self._menus[0].number_of_tasks
        if env_menu is not None:
            env_menu = np.ascontiguousarray(env_menu, dtype=np.uint8)
            if env_menu.shape != (self.num_envs,):
                raise ValueError('env_menu must have shape (num_envs,)')
        self._env_menu = env_menu

        cfg = L.cw_config()
        cfg.abi_version = L.CW_ABI_VERSION
        cfg.num_envs = self.num_envs
        cfg.size = self.size
        cfg.max_steps = self.MAX_STEPS
        cfg.n_task_list = len(self.task_list)
        cfg.fixed_init_state = self.fixed_init_state
        cfg.obs_mode = _OBS_MODES[obs_mode]
        cfg.auto_reset = 1 if self.auto_reset else 0
        cfg.keep_terminal_obs = 1 if (keep_terminal_obs and obs_mode != 'state') else 0
        cfg.raster = L.CW_RASTER_ALT if raster == 'alt' else L.CW_RASTER_RAY
        self.raster = raster
        self.host_outputs = bool(host_outputs)
        cfg.host_outputs = 1 if self.host_outputs else 0
        cfg.n_menus = len(menus)
        cfg.menus = self._menus
        cfg.env_menu = env_menu.ctypes.data_as(C.POINTER(C.c_uint8)) if env_menu is not None else None
        h_ = C.c_void_p()
        L.check(self._lib.cw_create(C.byref(cfg), self.device.index, C.byref(h_)), 'cw_create', self._lib)
        self._h = h_
        _LIVE.add(self)

        # zero-copy views of the engine's buffers
        tab = L.cw_buffer_table()
        L.check(self._lib.cw_buffers(self._h, C.byref(tab)), 'cw_buffers', self._lib)
        N, di = self.num_envs, self.device.index
        # frame geometry: ray.py:84 (4W,4H,3) / craftingworld_altobs.py:115 ((W+1)*3, H*3, 3)
        self.frame_shape = (3 * self.size + 3, 3 * self.size, 3) if raster == 'alt' else (4 * self.size, 4 * self.size, 3)
        fs = (N,) + self.frame_shape
        assert tab.frame_bytes == fs[1] * fs[2] * fs[3]
        dv = lambda p, shape, dt: tensor_view(p, shape, dt, di)  # noqa: E731
        v = (lambda p, shape, dt: _host_view(p, shape, dt)) if self.host_outputs else dv  # noqa: E731
        self._host_actions = _host_view(tab.host_actions, (N,), torch.int32).numpy() if self.host_outputs else None
        self._host_onehot = _host_view(tab.host_onehot, (self.size, self.size, 12), torch.uint8).numpy() if (self.host_outputs and tab.host_onehot) else None
        self._obs = v(tab.obs, fs, torch.uint8)
        self._desired_img = v(tab.desired_goal, fs, torch.uint8)
        self._init_img = v(tab.init_obs, fs, torch.uint8)
        self.terminal_observation = v(tab.terminal_obs, fs, torch.uint8)   # None unless keep_terminal_obs
        self.reward = v(tab.reward, (N,), torch.int32)
        self._done_u8 = v(tab.done, (N,), torch.uint8)
        self.done = self._done_u8.view(torch.bool)
        self.achieved_mask = v(tab.achieved, (N,), torch.int16)      # bit i = task_list[i] (bit pattern of a u16)
        self.desired_mask = v(tab.desired, (N,), torch.int16)
        self.episode_length = v(tab.episode_length, (N,), torch.int32)
        self.episode_return = v(tab.episode_return, (N,), torch.int32)   # sum of the finished episode's rewards (ray.py:361-367), valid where done
        self.hdr = dv(tab.hdr, (N, 16), torch.uint8)                 # packed current state, layout in craftingworld.h
        self.slot_pos = dv(tab.slot_pos, (N, 8), torch.int16)
        self.counters = dv(tab.counters, (4,), torch.int64)
        self._counters_raw = dv(tab.counters, (8,), torch.int64)     # (+ the engine's private words, craftingworld.h: tests only)
        self.agent_rc = self.hdr[:, 0:2]
        self.hold = self.hdr[:, 2]

        pix = self.frame_shape
        self.single_action_space = Discrete(len(ACTION_NAMES))       # ray.py:133
        self.action_space = MultiDiscrete([len(ACTION_NAMES)] * N)
        if obs_mode == 'state':
            self.single_observation_space = Dict(dict(hdr=Box(0, 255, (16,), np.uint8), slot_pos=Box(-2, 32767, (8,), np.int16)))
        else:
            self.single_observation_space = Dict({k: Box(0, 255, pix, np.uint8) for k in
                                                  ('observation', 'desired_goal', 'achieved_goal', 'init_observation')})
        self.observation_space = batch_space(self.single_observation_space, N)   # gym.vector: leading N on every Box
        self.observation_vector_space = Dict(dict(                   # ray.py:94-110
            observation=Box(0, 1, (self.size, self.size, 12), np.uint8),
            desired_goal=Box(0, 1, (1, len(self.task_list)), np.uint8),
            achieved_goal=Box(0, 1, (1, len(self.task_list)), np.uint8),
            init_observation=Box(0, 1, (self.size, self.size, 12), np.uint8)))
        self._pending = False
        self._actions_keepalive = None
        self._cw_step, self._di = self._lib.cw_step, self.device.index
        self._raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None) or (lambda di: torch.cuda.current_stream(di).cuda_stream)
        self.seed(seed)                                              # ray.py:70 (OS entropy when None)
        if self.fixed_init_state:                                    # ray.py:116-118
            L.check(self._lib.cw_generate_fixed_states(self._h, self._stream()), 'cw_generate_fixed_states', self._lib)

    # ------------------------------------------------------------------ plumbing
    def _stream(self):
        try:        # the raw handle without building a torch.cuda.Stream object (this sits on the per-step path)
            return C.c_void_p(torch._C._cuda_getCurrentRawStream(self.device.index))
        except AttributeError:
            return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def close_extras(self, **kwargs):
        """gym.vector hook: nothing besides the engine to release."""

    def close(self, **kwargs):
        if getattr(self, '_h', None):
            self.close_extras(**kwargs)
            self._lib.cw_destroy(self._h)
            self._h = None

    @property
    def closed(self):
        return getattr(self, '_h', None) is None

    @property
    def unwrapped(self):
        return self

    def __repr__(self):
        return 'CraftingWorldVecEnv(%d envs, %dx%d, obs_mode=%r, %s)' % (self.num_envs, self.size, self.size, self.obs_mode, self.device)

    def __del__(self):
        if sys is None or sys.is_finalizing():
            return
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    # ------------------------------------------------------------------ RNG (ray.py:145-147)
    def seed(self, seed=None):
        """gym.vector's seed(seeds): an int -> env i gets seed+i; a list/array of num_envs ints -> env i gets seeds[i];
        None -> OS entropy (ray.py:70).  seed_style 'numpy': env stream = numpy RandomState(seed_i);
        'gym': gym<=0.21 np_random(seed_i) hashing (seeding.py, unpinned).  Returns the per-env seeds."""
        if seed is not None and not isinstance(seed, (int, np.integer)):
            seeds = [seeding.create_seed(int(s)) for s in np.asarray(seed).reshape(-1)]
            if len(seeds) != self.num_envs:
                raise ValueError('expected %d seeds, got %d' % (self.num_envs, len(seeds)))
        else:
            base = seeding.create_seed(seed)
            seeds = [(base + i) for i in range(self.num_envs)]
        self._seeds = seeds
        if self.seed_style == 'gym':
            keys = np.empty((self.num_envs, L.CW_MT_N), dtype=np.uint32)
            pos = np.empty(self.num_envs, dtype=np.int32)
            for i, s in enumerate(seeds):
                keys[i], pos[i] = seeding.mt_state_from_seed(s)
            self.set_rng_states(keys, pos)
        else:
            arr = np.array([s & 0xFFFFFFFF for s in seeds], dtype=np.uint32)
            self._settle()
            L.check(self._lib.cw_seed_int(self._h, arr.ctypes.data_as(C.c_void_p)), 'cw_seed_int', self._lib)
        return seeds

    def set_rng_states(self, keys, pos):
        """Inject numpy RandomState states: keys uint32 [N,624], pos [N] (get_state()[1:3])."""
        keys = np.ascontiguousarray(keys, dtype=np.uint32)
        pos = np.ascontiguousarray(pos, dtype=np.int32)
        if keys.shape != (self.num_envs, L.CW_MT_N) or pos.shape != (self.num_envs,):
            raise ValueError('keys must be [N,624] and pos [N]')
        self._settle()
        L.check(self._lib.cw_seed_mt(self._h, keys.ctypes.data_as(C.c_void_p), pos.ctypes.data_as(C.c_void_p)), 'cw_seed_mt', self._lib)

    def get_rng_states(self):
        """-> (keys uint32 [N,624], pos int32 [N]) accepted by RandomState.set_state, same stream."""
        keys = np.empty((self.num_envs, L.CW_MT_N), dtype=np.uint32)
        pos = np.empty(self.num_envs, dtype=np.int32)
        self._settle()
        L.check(self._lib.cw_get_mt(self._h, keys.ctypes.data_as(C.c_void_p), pos.ctypes.data_as(C.c_void_p)), 'cw_get_mt', self._lib)
        return keys, pos

    # ------------------------------------------------------------------ gym.vector surface
    def _observation(self):
        if self.obs_mode == 'state':
            return {'hdr': self.hdr, 'slot_pos': self.slot_pos}
        return {'observation': self._obs, 'desired_goal': self._desired_img, 'achieved_goal': self._obs,
                'init_observation': self._init_img}                  # achieved_goal IS observation, ray.py:194-196

    def _sync(self):
        L.check(self._lib.cw_synchronize(self._h, self._stream()), 'cw_synchronize', self._lib)

    def _settle(self):
        """Ahead of a synchronous library call (seeding, state get/set, checkpoints): wait for torch's CURRENT stream.  The library waits for the
        streams the engine was handed by itself -- but a replayed HIP graph runs wherever torch replays it, which the engine never sees."""
        if getattr(self, '_h', None):
            self._lib.cw_synchronize(self._h, self._stream())

    def synchronize(self):
        """Block until everything this env enqueued on the current stream (and on its own side stream) has finished."""
        self._sync()

    def reset_async(self):
        L.check(self._lib.cw_reset(self._h, self._stream()), 'cw_reset', self._lib)     # enqueues; nothing waits
        self._has_reset = True
        self._reset_pending = True

    def reset_wait(self):
        if not getattr(self, '_reset_pending', False):
            raise RuntimeError('reset_wait without reset_async')
        self._reset_pending = False
        if self.host_outputs:
            self._sync()
        return self._observation()

    def reset(self):
        self.reset_async()
        return self.reset_wait()

    def step_async(self, actions):
        # the per-step path: a device tensor of the right shape goes to cw_step with nothing built on the way (pointer and stream as plain ints)
        if type(actions) is torch.Tensor and actions.is_cuda:
            dt = _ACT_DTYPES.get(actions.dtype)
            if dt is not None and actions.numel() == self.num_envs and actions.is_contiguous() and actions.device == self.device:
                self._actions_keepalive = actions
                rc = self._cw_step(self._h, actions.data_ptr(), dt, self._raw_stream(self._di))
                if rc:
                    L.check(rc, 'cw_step', self._lib)
                self._pending = True
                return
        if self.host_outputs and not (torch.is_tensor(actions) and actions.is_cuda):
            a = np.asarray(actions).reshape(-1)             # host actions go through the engine's mapped buffer
            if a.size != self.num_envs:
                raise ValueError('expected %d actions, got %d' % (self.num_envs, a.size))
            self._host_actions[:] = a
            L.check(self._lib.cw_step(self._h, C.c_void_p(self._host_actions.ctypes.data), L.CW_ACT_I32, self._stream()),
                    'cw_step', self._lib)
            self._pending = True
            return
        if not torch.is_tensor(actions):
            actions = torch.as_tensor(np.asarray(actions), device=self.device)
        if actions.device != self.device:
            actions = actions.to(self.device)
        if actions.dtype not in _ACT_DTYPES:
            actions = actions.to(torch.int32)
        actions = actions.contiguous()
        if actions.numel() != self.num_envs:
            raise ValueError('expected %d actions, got %d' % (self.num_envs, actions.numel()))
        self._actions_keepalive = actions
        L.check(self._lib.cw_step(self._h, C.c_void_p(actions.data_ptr()), _ACT_DTYPES[actions.dtype], self._stream()),
                'cw_step', self._lib)
        self._pending = True

    def step_wait(self):
        if not self._pending:
            raise RuntimeError('step_wait without step_async')
        self._pending = False
        if self.host_outputs:
            self._sync()
        # 'episode': the finished episode's statistics the way gym's RecordEpisodeStatistics reports them (r: return = the sum of its rewards,
        # ray.py:361-367; l: length), as device tensors, rows valid where done (SURVEY 5 "metrics")
        info = {'task_success': self.achieved_mask, 'desired_goal': self.desired_mask,
                'achieved_goal': self.achieved_mask, 'episode_length': self.episode_length,
                'episode': {'r': self.episode_return, 'l': self.episode_length}}
        if self.terminal_observation is not None:
            info['terminal_observation'] = self.terminal_observation     # rows valid where done
        return self._observation(), self.reward, self.done, info

    def step(self, actions):
        """-> (obs, reward int32[N], done bool[N], info); all device tensors, results are ordered on the
        current torch stream (no host sync).  Finished envs are already reset: obs rows of done
        envs show the first frame of the new episode; info masks are the terminal ones."""
        self.step_async(actions)
        return self.step_wait()

    def step_many(self, actions):
        """K consecutive steps from a device tensor `actions` [K, N] (uint8 / int32 / int64): exactly what K calls of step_async enqueue, in ONE
        library call -- a Python loop costs more per step than a state-only step takes on the card.  Nothing is returned: read the usual
        views (reward, done, the observation tensors: the last step's) afterwards.  Auto-reset as in step()."""
        if not (type(actions) is torch.Tensor and actions.dtype in _ACT_DTYPES and actions.device == self.device and actions.is_contiguous()
                and actions.dim() == 2 and actions.shape[1] == self.num_envs):
            raise ValueError('actions must be a contiguous [K, num_envs] tensor of dtype uint8 / int32 / int64 on %s' % (self.device,))
        self._actions_keepalive = actions
        L.check(self._lib.cw_step_many(self._h, C.c_void_p(actions.data_ptr()), _ACT_DTYPES[actions.dtype], int(actions.shape[0]), self._stream()),
                'cw_step_many', self._lib)

    def capture_steps(self, actions):
        """-> a torch.cuda.CUDAGraph that takes K = actions.shape[0] steps per replay(), reading row t of the device tensor `actions` [K, N] on
        step t: fill the tensor IN PLACE (the ring a policy or an action sampler writes into), call graph.replay(), read the views.  One graph
        launch per K steps instead of K library calls: the way to run the launch-bound modes (state-only, dirty-cell frames) at the card's pace
        rather than the host's.  The envs' episode bookkeeping needs nothing from the host, so replays can be queued back to back.
        Capturing takes NO step: nothing runs until the first replay() (cw_step_many allocates nothing and launches nothing lazily, so torch's
        eager warm-up pass is not needed -- it would advance every env by K steps on whatever the tensor holds at that moment)."""
        self.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            self.step_many(actions)
        return graph

    def rollout(self, actions, record=True):
        """T consecutive steps (auto-reset included) in ONE persistent kernel launch, for action streams
        known up front: actions uint8 [T, N] on the device.  Bit-identical to T calls of step().
        State-only observation mode.  -> (rewards int32 [T,N], dones bool [T,N]) if record."""
        if not torch.is_tensor(actions):
            actions = torch.as_tensor(np.asarray(actions), device=self.device)
        actions = actions.to(device=self.device, dtype=torch.uint8).contiguous()
        if actions.dim() != 2 or actions.shape[1] != self.num_envs:
            raise ValueError('actions must have shape [T, num_envs]')
        T = actions.shape[0]
        rew = torch.empty((T, self.num_envs), dtype=torch.int32, device=self.device) if record else None
        don = torch.empty((T, self.num_envs), dtype=torch.uint8, device=self.device) if record else None
        L.check(self._lib.cw_rollout(self._h, C.c_void_p(actions.data_ptr()), T,
                                     C.c_void_p(rew.data_ptr()) if record else None,
                                     C.c_void_p(don.data_ptr()) if record else None, self._stream()), 'cw_rollout', self._lib)
        self._actions_keepalive = actions
        return (rew, don.view(torch.bool)) if record else None

    # ------------------------------------------------------------------ views / checkpoints
    def render(self, out=None):
        """render() of ray.py:442-520 for every env -> uint8 [N,4S,4S,3] (works in every obs_mode)."""
        if out is None:
            out = torch.empty((self.num_envs,) + self.frame_shape, dtype=torch.uint8, device=self.device)
        L.check(self._lib.cw_render(self._h, C.c_void_p(out.data_ptr()), self._stream()), 'cw_render', self._lib)
        return out

    def render_exact(self):
        """The reference's INT image of every env's current state, exactly: int16 [N, ...frame_shape].  The uint8 frames of the engine hold it
        modulo 256, which differs in ONE reachable pixel of the AltObs raster (sticks held over a sticks cell: (90, 164, 320),
        craftingworld_altobs.py:527-543) and nowhere in the Ray raster; this is the batch counterpart of the N=1 classes' reference_dtypes=True.
        Two kernels (cw_export_onehot, cw_render_onehot): a side view for checks and logging, not a per-step observation."""
        return self.render_states(self.one_hot())

    def render_states(self, one_hot):
        """render(state) of ray.py:442-486 for caller-supplied one-hot states [M,S,S,12] of any content (several objects in a
        cell, any number of objects): -> int16 tensor [M,4S,4S,3] holding the reference's int image (sums of colours; the
        agent is the first cell with channel 8 set and must exist).  raster='alt': CraftingWorldEnvAltObs.render(state)
        (craftingworld_altobs.py:489-560), [M,3S+3,3S,3].  Does not touch the envs."""
        oh = torch.as_tensor(np.asarray(one_hot) if not torch.is_tensor(one_hot) else one_hot)
        oh = oh.to(device=self.device, dtype=torch.uint8).contiguous()
        if oh.dim() != 4 or tuple(oh.shape[1:]) != (self.size, self.size, 12):
            raise ValueError('states must have shape [M, %d, %d, 12]' % (self.size, self.size))
        out = torch.empty((oh.shape[0],) + self.frame_shape, dtype=torch.int16, device=self.device)
        L.check(self._lib.cw_render_onehot(self._h, C.c_void_p(oh.data_ptr()), oh.shape[0], C.c_void_p(out.data_ptr()), self._stream()),
                'cw_render_onehot', self._lib)
        self._states_keepalive = oh
        return out

    def grid(self, out=None):
        """Dense cell codes uint8 [N,S,S] (0 empty, k+1 = OBJECTS[k])."""
        if out is None:
            out = torch.empty((self.num_envs, self.size, self.size), dtype=torch.uint8, device=self.device)
        L.check(self._lib.cw_export_grid(self._h, C.c_void_p(out.data_ptr()), self._stream()), 'cw_export_grid', self._lib)
        return out

    def one_hot(self, out=None, which='current'):
        """obs_one_hot of ray.py:119 for every env: uint8 [N,S,S,12].  which='goal' / 'init' give the episode's goal state
        and its state at reset -- CraftingWorldEnvOneHot's desired_goal and init_observation (onehot.py:310, :203)."""
        if out is None:
            out = torch.empty((self.num_envs, self.size, self.size, 12), dtype=torch.uint8, device=self.device)
        w = {'current': 0, 'goal': 1, 'init': 2}[which]
        L.check(self._lib.cw_export_onehot_of(self._h, w, C.c_void_p(out.data_ptr()), self._stream()), 'cw_export_onehot_of', self._lib)
        return out

    def mask_to_vector(self, mask):
        """int16 bit mask [N] -> 0/1 tensor [N, len(task_list)] (achieved_goal_vector layout, ray.py:114)."""
        bits = torch.arange(len(self.task_list), device=mask.device, dtype=torch.int32)
        return ((mask.to(torch.int32).unsqueeze(1) >> bits) & 1).to(torch.uint8)

    def fixed_states(self):
        """generate_fixed_states' pool (ray.py:116-118, 149-154: fixed_state_list) as cell indices: uint16 [N, K, 9] = the cells (row * S + col)
        of objects 0..7 (OBJECTS order) and of the agent in each of the K pooled placements of every env.  ValueError when fixed_init_state == 0."""
        out = np.empty((self.num_envs, max(self.fixed_init_state, 1), 9), dtype=np.uint16)
        self._settle()
        L.check(self._lib.cw_get_fixed_states(self._h, out.ctypes.data_as(C.c_void_p)), 'cw_get_fixed_states', self._lib)
        return out

    def get_state(self):
        """Host snapshot (numpy) of every env: the de-facto checkpoint."""
        N, S = self.num_envs, self.size
        out = dict(grid=np.empty((N, S, S), np.uint8), init_grid=np.empty((N, S, S), np.uint8),
                   goal_grid=np.empty((N, S, S), np.uint8), agent_rc=np.empty((N, 2), np.uint8),
                   init_agent_rc=np.empty((N, 2), np.uint8), goal_agent_rc=np.empty((N, 2), np.uint8),
                   hold=np.empty(N, np.uint8), achieved=np.empty(N, np.uint16), desired=np.empty(N, np.uint16),
                   step_num=np.empty(N, np.int32), ep_no=np.empty(N, np.int32))
        view = L.cw_state_view(**{k: a.ctypes.data_as(C.c_void_p) for k, a in out.items()})
        self._settle()
        L.check(self._lib.cw_get_state(self._h, C.byref(view)), 'cw_get_state', self._lib)
        return out

    def set_state(self, **fields):
        """Overwrite state from numpy arrays: any subset of get_state()'s fields (grid, init_grid, goal_grid, agent_rc,
        init_agent_rc, goal_agent_rc, hold, achieved, desired, step_num, ep_no); with a goal / init-agent field given,
        all three frames are repainted from the restored states."""
        dt = dict(grid=np.uint8, init_grid=np.uint8, goal_grid=np.uint8, agent_rc=np.uint8, init_agent_rc=np.uint8,
                  goal_agent_rc=np.uint8, hold=np.uint8, achieved=np.uint16, desired=np.uint16, step_num=np.int32,
                  ep_no=np.int32)
        keep = {}
        view = L.cw_state_view()
        for k, a in fields.items():
            if k not in dt:
                raise ValueError('cannot set %r' % k)
            keep[k] = np.ascontiguousarray(a, dtype=dt[k])
            if keep[k].shape[0] != self.num_envs:
                raise ValueError('%s must have num_envs rows' % k)
            setattr(view, k, keep[k].ctypes.data_as(C.c_void_p))
        self._settle()
        L.check(self._lib.cw_set_state(self._h, C.byref(view)), 'cw_set_state', self._lib)

    # ------------------------------------------------------------------ checkpoint / resume
    def save_checkpoint(self, path):
        """Everything the env batch needs to continue bit-identically, as one file written exactly at `path`: the engine's
        raw records (current state, the episode's goal and start states, RNG streams, fixed_init_state pools, the last
        step's outputs, counters) behind a header that pins the configuration (cw_checkpoint_save; synchronises)."""
        n = int(self._lib.cw_checkpoint_bytes(self._h))
        buf = np.empty(n, dtype=np.uint8)
        self._settle()
        L.check(self._lib.cw_checkpoint_save(self._h, buf.ctypes.data_as(C.c_void_p), n), 'cw_checkpoint_save', self._lib)
        with open(path, 'wb') as f:
            buf.tofile(f)

    def load_checkpoint(self, path):
        """Restore a save_checkpoint() file into an engine built with the same configuration (num_envs, size, max_steps,
        task_list, fixed_init_state and task menus are verified: ValueError otherwise).  Records are restored verbatim, so
        the state-mode observation tensors (hdr, slot_pos) equal the uninterrupted run's too; frames are repainted.  Per-env
        menu ids, reward rules, pools and counters come from the file."""
        buf = np.fromfile(path, dtype=np.uint8)
        self._settle()
        L.check(self._lib.cw_checkpoint_load(self._h, buf.ctypes.data_as(C.c_void_p), buf.size), 'cw_checkpoint_load', self._lib)
        self._has_reset = True

    def profile_begin(self, max_steps):
        """Bracket each kernel of the following step() calls with HIP events on the launch stream."""
        L.check(self._lib.cw_profile_begin(self._h, int(max_steps)), 'cw_profile_begin', self._lib)

    def render_kernel_name(self):
        """Name of the kernel `profile_end()['ms_render_kernel']` brackets (as a rocprofv3 kernel trace lists it)."""
        return (self._lib.cw_render_kernel_name(self._h) or b'').decode()

    def profile_end(self):
        """-> dict of average per-launch kernel durations (ms) since profile_begin."""
        p = L.cw_profile()
        L.check(self._lib.cw_profile_end(self._h, C.byref(p)), 'cw_profile_end', self._lib)
        return {k: getattr(p, k) for k, _ in p._fields_}

    def tuner_state(self):
        """What the engine's tuning holds right now (full-frame mode; performance only): dict with `period16` (the period of the sweep's clock in
        1/16 of a 10-ns tick, 0: unclocked; `period16_head`: of a launch's first 64 jobs, `period16_busy`: of those after a step on which envs finished), `lookahead` (1: the outcome of every env's next reset() is computed ahead of time), `resident` (1: the
        N=1 doorbell stepper is available) and `guard_slowdowns` (how often the clock's guard has lowered the rate; -1: no guard)."""
        t = L.cw_tuner_state()
        L.check(self._lib.cw_tuner(self._h, C.byref(t)), 'cw_tuner', self._lib)
        return {k: int(getattr(t, k)) for k, _ in t._fields_}

    def compute_reward_batch(self, achieved_mask, desired_mask, subset=None):
        """Vectorised compute_reward_equal / compute_reward_subset (ray.py:757-767) on bit masks, for
        HER-style relabelling on the device: int tensors of any shape -> int32 rewards (MAX_STEPS or -1).
        subset=None uses menu 0's reward_style."""
        full = (1 << len(self.task_list)) - 1
        a = achieved_mask.to(torch.int32) & full
        d = desired_mask.to(torch.int32) & full
        if subset is None:
            subset = bool(self._menus[0].reward_subset)
        if subset:      # np.max(desired - achieved) == 0: nothing missing and at least one position equal
            hit = ((d & ~a) == 0) & (((~(d ^ a)) & full) != 0)
        else:
            hit = a == d
        return torch.where(hit, torch.full_like(a, self.MAX_STEPS), torch.full_like(a, -1))

    def compute_reward(self, achieved_goal, desired_goal, info=None):
        """compute_reward_equal / compute_reward_subset of ray.py:757-767 on 0/1 goal vectors
        (host convenience for HER-style relabelling; the per-step reward comes from the kernel)."""
        a = np.asarray(achieved_goal).reshape(-1)
        d = np.asarray(desired_goal).reshape(-1)
        if self._menus[0].reward_subset:
            return self.MAX_STEPS if np.max(d.astype(np.int64) - a.astype(np.int64)) == 0 else -1
        return self.MAX_STEPS if np.array_equal(a, d) else -1
