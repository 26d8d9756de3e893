"""A stand-in for CraftingWorldVecEnv(1, host_outputs=True) built on the CPU oracle -- TEST INFRASTRUCTURE ONLY.

The N=1 facade classes (gym_craftingworld_amd/env.py) are host logic on top of the engine: which arrays are handed out and when they are
rebound, np_random as a live generator, step_num / ep_no bookkeeping, negative action ids, the reference's exceptions.  None of that needs a GPU
to be WRONG, so the CPU tier runs it: `install(monkeypatch)` swaps the engine class the facade constructs for FakeVecEnv below, which implements
exactly the attributes and library entry points env.py touches, with the oracle as the thing that steps.  The product never sees this file; the
GPU tier replays the same fixtures through the real engine (tests/test_hip_parity.py).
"""
import ctypes as C

import numpy as np
import torch

from oracle import OracleEnv
from gym_craftingworld_amd import seeding
from gym_craftingworld_amd.spaces import Box, Dict

TASK_LIST = ['MakeBread', 'EatBread', 'BuildHouse', 'ChopTree', 'ChopRock', 'GoToHouse', 'MoveAxe', 'MoveHammer', 'MoveSticks']


def _one_hot(grid, agent, hold=0):
    oh = np.zeros(grid.shape + (12,), dtype=np.uint8)
    r, c = np.nonzero(grid)
    oh[r, c, grid[r, c] - 1] = 1
    if agent[0] >= 0:
        oh[agent[0], agent[1], 8] = 1
        if hold:
            oh[agent[0], agent[1], 8 + hold] = 1
    return oh


class _FakeLib:
    """the library entry points env.py calls directly (everything else goes through the vec-env methods)"""

    def __init__(self, vec):
        self._v = vec

    def _into(self, ptr, arr):
        n = arr.size
        dst = np.frombuffer((C.c_uint8 * n).from_address(ptr.value if isinstance(ptr, C.c_void_p) else int(ptr)), dtype=np.uint8)
        dst[:] = np.ascontiguousarray(arr, np.uint8).reshape(-1)

    def cw_step(self, eng, act_p, dtype, stream):
        return self._v._do_step(int(self._v._host_actions[0]))

    def cw_step_resident(self, eng, action, want_onehot):
        rc = self._v._do_step(int(action))
        if rc == 0 and want_onehot:
            self._v._host_onehot[...] = self._v._current_onehot()
        return rc

    def cw_synchronize(self, eng, stream):
        return 0

    def cw_generate_fixed_states(self, eng, stream):
        self._v._ora._lib.cwo_generate_fixed_states(self._v._ora._h)
        return 0

    def cw_export_onehot(self, eng, ptr, stream):
        self._into(ptr, self._v._current_onehot())
        return 0

    def cw_export_onehot_of(self, eng, which, ptr, stream):
        s = self._v._ora.state()
        oh = (self._v._current_onehot() if which == 0 else _one_hot(s['goal_grid'], s['goal_agent']) if which == 1 else
              _one_hot(s['init_grid'], s['init_agent']))
        self._into(ptr, oh)
        return 0

    def cw_last_error(self):
        return b'fake engine'


class FakeVecEnv:
    """CraftingWorldVecEnv(1, obs_mode='pixels_dirty', auto_reset=False, host_outputs=True, seed_style='gym') as env.py uses it"""
    resident = True        # class switch: tests run both step paths of the facade

    def __init__(self, num_envs, size=(21, 21), fixed_init_state=0, max_steps=300, store_gif=False, render_save_rate=1, task_list=TASK_LIST,
                 selected_tasks=TASK_LIST, number_of_tasks=None, stacking=True, reward_style=None, obs_mode='pixels_dirty', device=None, seed=None,
                 seed_style='gym', auto_reset=False, raster='ray', host_outputs=True, **kw):
        assert num_envs == 1 and host_outputs and not auto_reset and obs_mode == 'pixels_dirty' and seed_style == 'gym'
        if size[0] != size[1]:
            raise ValueError('non-square grids are rejected')
        self._kw = dict(size=tuple(size), fixed_init_state=fixed_init_state, max_steps=max_steps, task_list=list(task_list),
                        selected_tasks=list(selected_tasks), number_of_tasks=number_of_tasks, stacking=stacking, reward_style=reward_style,
                        alt_obs=(raster == 'alt'))
        self.num_envs, self.size = 1, size[0]
        self.STATE_W = self.STATE_H = size[0]
        self.MAX_STEPS, self.task_list, self.selected_tasks = max_steps, list(task_list), list(selected_tasks)
        n = number_of_tasks if number_of_tasks is not None else len(self.selected_tasks)
        self.number_of_tasks = min(n, len(self.selected_tasks))
        self.fixed_init_state = fixed_init_state
        # the oracle is created WITHOUT drawing its fixed-state pool: the constructor below seeds first and draws the pool afterwards, like the engine
        self._ora = _with_pool(self._kw)
        self.frame_shape = self._ora.img_shape
        S, fs = self.size, self.frame_shape
        t = lambda shape, dt: torch.zeros(shape, dtype=dt)  # noqa: E731
        self._obs, self._desired_img, self._init_img = t((1,) + fs, torch.uint8), t((1,) + fs, torch.uint8), t((1,) + fs, torch.uint8)
        self.reward, self._done_u8 = t((1,), torch.int32), t((1,), torch.uint8)
        self.achieved_mask, self.desired_mask = t((1,), torch.int16), t((1,), torch.int16)
        self._host_actions = np.zeros(1, np.int32)
        self._host_onehot = np.zeros((S, S, 12), np.uint8)
        self.observation_vector_space = Dict(dict(observation=Box(0, 1, (S, S, 12), np.uint8), desired_goal=Box(0, 1, (1, len(self.task_list)), np.uint8),
                                                  achieved_goal=Box(0, 1, (1, len(self.task_list)), np.uint8), init_observation=Box(0, 1, (S, S, 12), np.uint8)))
        self._lib, self._h = _FakeLib(self), C.c_void_p(1)
        self._has_reset = False
        self._ep_no_offset = 0
        self.n_rng_uploads = self.n_rng_downloads = 0       # (the tests count the mirror's traffic)
        self.seed(seed)
        if fixed_init_state:
            self._ora._lib.cwo_generate_fixed_states(self._ora._h)

    # ---- plumbing the facade reads
    def _stream(self):
        return C.c_void_p(0)

    def tuner_state(self):
        return {'resident': 1 if type(self).resident else 0}

    @property
    def agent_rc(self):
        s = self._ora.state()
        return torch.tensor([[s['agent'][0], s['agent'][1]]], dtype=torch.uint8)

    def _current_onehot(self):
        s = self._ora.state()
        return _one_hot(s['grid'], s['agent'], s['hold'])

    def _publish(self, frames=True):
        v = self._ora.view()
        ish = self.frame_shape
        self._obs[0].numpy()[...] = self._ora._arr(v.obs, ish)
        if frames:
            self._desired_img[0].numpy()[...] = self._ora._arr(v.desired_img, ish)
            self._init_img[0].numpy()[...] = self._ora._arr(v.init_img, ish)
        self.achieved_mask.numpy().view(np.uint16)[0] = v.achieved
        self.desired_mask.numpy().view(np.uint16)[0] = v.desired

    def _do_step(self, a):
        if not self._has_reset:
            return -3
        if not 0 <= a < 6:
            a = 6                                    # (the engine's counted no-op; the facade never sends one)
        r, d = C.c_int32(), C.c_int32()
        if self._ora._lib.cwo_step(self._ora._h, a, C.byref(r), C.byref(d)) != 0:
            return -1
        self.reward[0], self._done_u8[0] = r.value, d.value
        self._publish(frames=False)
        return 0

    # ---- RNG
    def seed(self, seed=None):
        s = seeding.create_seed(seed)
        key, pos = seeding.mt_state_from_seed(s)
        self.set_rng_states(key[None], np.array([pos]))
        self._seeds = [s]
        return [s]

    def set_rng_states(self, keys, pos):
        self.n_rng_uploads += 1
        self._ora.set_rng(np.asarray(keys, np.uint32)[0], int(np.asarray(pos)[0]))

    def get_rng_states(self):
        self.n_rng_downloads += 1
        k, p = self._ora.get_rng()
        return k[None].copy(), np.array([p], np.int32)

    # ---- env
    def reset(self):
        self._ora.reset()
        self._has_reset = True
        self._publish()
        return None

    def get_state(self):
        s = self._ora.state()
        u8 = lambda a: np.asarray(a, np.uint8)[None]  # noqa: E731
        return dict(grid=u8(s['grid']), init_grid=u8(s['init_grid']), goal_grid=u8(s['goal_grid']), agent_rc=u8(s['agent']),
                    init_agent_rc=u8(np.maximum(s['init_agent'], 0)), goal_agent_rc=u8(s['goal_agent']), hold=np.array([s['hold']], np.uint8),
                    achieved=np.array([s['achieved']], np.uint16), desired=np.array([s['desired']], np.uint16),
                    step_num=np.array([s['step_num']], np.int32), ep_no=np.array([s['ep_no'] + self._ep_no_offset], np.int32))

    def set_state(self, **f):
        s = self._ora.state()
        if 'ep_no' in f:
            self._ep_no_offset = int(np.asarray(f.pop('ep_no')).reshape(-1)[0]) - s['ep_no']
        for k in f:
            if k not in ('grid', 'init_grid', 'agent_rc', 'hold', 'achieved', 'desired', 'step_num'):
                raise NotImplementedError('the fake engine cannot restore %r' % k)
        if f:
            g = lambda k, d: np.asarray(f[k])[0] if k in f else d  # noqa: E731
            self._ora.set_state(g('grid', s['grid']), g('init_grid', s['init_grid']), tuple(int(x) for x in g('agent_rc', s['agent'])),
                                int(g('hold', s['hold'])), int(g('achieved', s['achieved'])), int(g('desired', s['desired'])), int(g('step_num', s['step_num'])))
            self._publish(frames=False)

    def fixed_states(self):
        if not self.fixed_init_state:
            raise ValueError('fixed_init_state == 0')
        return OracleEnv.fixed_states(self._ora)[None]

    def render(self):
        return self._obs.clone()

    def compute_reward(self, achieved_goal, desired_goal, info=None):
        a, d = np.asarray(achieved_goal).reshape(-1), np.asarray(desired_goal).reshape(-1)
        if self._kw['reward_style'] is not None:
            return self.MAX_STEPS if np.max(d.astype(np.int64) - a.astype(np.int64)) == 0 else -1
        return self.MAX_STEPS if np.array_equal(a, d) else -1

    def close(self):
        self._ora = None


def _with_pool(kw):
    """an OracleEnv whose fixed-state pool is drawn LATER (the facade seeds first): OracleEnv's constructor draws it at once, from the default stream"""
    cfg_kw = dict(kw)
    env = OracleEnv.__new__(OracleEnv)
    from oracle.oracle import _lib, make_config
    env._lib = _lib()
    env.cfg = make_config(**cfg_kw)
    env._h = env._lib.cwo_new(C.byref(env.cfg))
    if not env._h:
        raise ValueError('cwo_new rejected the config')
    env.size = env.cfg.size
    env.img_shape = ((3 * env.cfg.size + 3, 3 * env.cfg.size, 3) if env.cfg.alt_obs else (4 * env.cfg.size, 4 * env.cfg.size, 3))
    env.MAX_STEPS, env.n_task_list = env.cfg.max_steps, env.cfg.n_task_list
    return env


def install(monkeypatch=None, resident=True):
    """make gym_craftingworld_amd.env construct FakeVecEnv instead of the HIP engine (CPU tier only; monkeypatch=None: set for the rest of the
    process -- tools/diff_vs_reference.py, in the build container)"""
    import gym_craftingworld_amd.env as E
    cls = type('FakeVecEnv_%s' % ('resident' if resident else 'launch'), (FakeVecEnv,), {'resident': resident})
    put = monkeypatch.setattr if monkeypatch is not None else setattr
    put(E, 'CraftingWorldVecEnv', cls)
    put(E, '_pinned_u8', lambda shape: torch.zeros(shape, dtype=torch.uint8))       # (no GPU here: plain host memory)
    return cls
