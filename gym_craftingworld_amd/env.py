"""Single-env gym.Env-shaped façades over the HIP engine with num_envs=1 -- the drop-in for
`gym.make('craftingworld-v3')` & co. (reference registration: gym_craftingworld/__init__.py:5-18).

The engine runs with cw_config.host_outputs: its kernels write frames, reward, done and the goal masks straight
into pinned host memory, so a step() is one kernel launch and one stream sync -- no copies.  With the default
uint8 dtype the returned arrays ARE those buffers, mutated in place by later steps: the reference's own aliasing
contract (obs['observation'] is env.obs_image, ray.py:194-196).

Same ctor kwargs, return shapes and aliasing as the reference classes:
  CraftingWorldEnv        <- CraftingWorldEnvRay      (craftingworld_ray.py:53)
  CraftingWorldEnvFlat    <- CraftingWorldEnvFlat     (craftingworld_flat.py:46)
  CraftingWorldEnvOneHot  <- CraftingWorldEnvOneHot   (carftingworld_onehot.py:53)
Host arrays are numpy (uint8 by default; reference_dtypes=True up-casts to the reference's int64).
No auto-reset: after done the caller calls reset(), exactly like the reference loop
(docs/source/envs/gen_info.rst:62-82).  Compute still runs on the GPU; there is no CPU path.
"""
import ctypes as C

import numpy as np
import torch

from . import seeding
from .coord import GridPos
from .spaces import Box, Dict, Discrete
from .vec_env import ACTION_NAMES, TASK_LIST, CraftingWorldVecEnv


class CraftingWorldEnv:
    metadata = {'render.modes': ['human', 'Non']}
    reward_range = (-float('inf'), float('inf'))       # gym.Env's class attributes (the reference inherits them from gym.GoalEnv, ray.py:53): wrappers read them
    spec = None
    _default_size = (21, 21)
    _default_max_steps = 300

    _raster = 'ray'

    def __init__(self, size=None, fixed_init_state=0, max_steps=None, store_gif=False, render_save_rate=1,
                 task_list=TASK_LIST, selected_tasks=TASK_LIST, number_of_tasks=None, stacking=True,
                 reward_style=None, device=None, reference_dtypes=False, seed=None, resident=None):
        size = self._default_size if size is None else size
        max_steps = self._default_max_steps if max_steps is None else max_steps
        self._vec = CraftingWorldVecEnv(1, size=size, fixed_init_state=fixed_init_state, max_steps=max_steps,
                                        store_gif=False, render_save_rate=render_save_rate, task_list=task_list,
                                        selected_tasks=selected_tasks, number_of_tasks=number_of_tasks,
                                        stacking=stacking, reward_style=reward_style, obs_mode='pixels_dirty',
                                        device=device, seed=seed, seed_style='gym', auto_reset=False,
                                        raster=self._raster, host_outputs=True)
        v = self._vec
        self.STATE_W, self.STATE_H = v.STATE_W, v.STATE_H
        self.MAX_STEPS = v.MAX_STEPS
        self.task_list, self.selected_tasks = v.task_list, v.selected_tasks
        self.number_of_tasks, self.stacking = v.number_of_tasks, stacking
        self.fixed_init_state = fixed_init_state
        self._dtype = np.int64 if reference_dtypes else np.uint8
        fs = v.frame_shape
        img = lambda: Box(low=0, high=255, shape=fs, dtype=self._dtype)  # noqa: E731
        self.observation_space = Dict(dict(observation=img(), desired_goal=img(), achieved_goal=img(),
                                           init_observation=img()))                # ray.py:85-92
        self.observation_vector_space = v.observation_vector_space                # ray.py:94-110
        self.action_space = Discrete(len(ACTION_NAMES))                            # ray.py:133
        # ray.py:130-131: the four moves are Coord offsets carrying their name, pickup / drop plain strings
        self.ACTIONS = [GridPos(-1, 0, name='up'), GridPos(0, 1, name='right'), GridPos(1, 0, name='down'), GridPos(0, -1, name='left'),
                        'pickup', 'drop']
        assert [getattr(a, 'name', a) for a in self.ACTIONS] == list(ACTION_NAMES)
        self.ep_no = 0
        self.step_num = 0
        self._has_reset = False                     # (ray.py:119-124: obs_one_hot, agent_pos, INIT_OBS_VECTOR, observation_vector are None until reset())
        # the engine's host-mapped buffers (row 0 of the batch of one)
        self._h_obs, self._h_goal, self._h_init = (t[0].numpy() for t in (v._obs, v._desired_img, v._init_img))
        self._h_reward, self._h_done = v.reward.numpy(), v._done_u8.numpy()
        self._h_ach, self._h_des = v.achieved_mask.numpy().view(np.uint16), v.desired_mask.numpy().view(np.uint16)
        self._live = not reference_dtypes           # uint8: hand out the live buffers themselves
        self.obs_image = self._h_obs if self._live else np.zeros(fs, self._dtype)
        self.desired_goal = self._h_goal if self._live else np.zeros(fs, self._dtype)
        self.INIT_OBS = self._h_init if self._live else np.zeros(fs, self._dtype)
        self.observation = None
        self.desired_goal_vector = np.zeros((1, len(self.task_list)), dtype=int)  # ray.py:112
        self._masks_seen = (-1, -1)                 # the (achieved, desired) masks the two vectors currently show
        self.achieved_goal_vector = np.zeros((1, len(self.task_list)), dtype=int)
        self._bit_rows = ((np.arange(1 << len(self.task_list))[:, None] >> np.arange(len(self.task_list))) & 1) \
            if len(self.task_list) <= 12 else None                                # mask -> 0/1 row
        # the per-step call sequence, bound once: cw_step on the mapped action buffer, then one stream sync
        self._lib, self._eng, self._stream = v._lib, v._h, v._stream()
        self._act = v._host_actions
        self._act_p = C.c_void_p(self._act.ctypes.data)
        # step() without a kernel launch: a resident single-wave kernel polls a doorbell in pinned host memory (cw_step_resident; it leaves by
        # itself after 0.5 ms without a request and is parked by every other call).  resident=False (or CW_RESIDENT=0) keeps "launch + stream sync".
        import os
        self._resident = (os.environ.get('CW_RESIDENT', '1') != '0') if resident is None else bool(resident)
        self._resident = self._resident and bool(v.tuner_state()['resident'])     # (no pinned control block / stream at cw_create: the launch path)
        self._step_resident = self._lib.cw_step_resident
        self._want_onehot = 0                       # (the one-hot class: the resident step also leaves obs_one_hot in pinned host memory)
        self.store_gif, self.render_save_rate = False, render_save_rate            # ray.py:135-136
        self._gif_frames = None
        if store_gif:                                                              # ray.py:142-143
            self.allow_gif_storage()

    # -- episode GIFs (ray.py:160-167, 205-216, 370-374, 769-782): host-side debug I/O off the step path.  Same files in
    # the same places (renders/env<id>/E<ep>(<steps>)_<desired>(<achieved>).gif, every render_save_rate-th episode) and
    # the same draw from the env's RNG stream for <id>; frames are observation | desired_goal side by side (pillow), not
    # the reference's matplotlib figure with legend and captions.
    def allow_gif_storage(self, store_gif=True):
        self.store_gif = store_gif
        if self.store_gif is True:
            import os
            rs = np.random.RandomState()
            k, p = self.get_rng_state()
            rs.set_state(('MT19937', k, p, 0, 0.0))
            self.env_id = int(rs.randint(0, 1000000))                              # np_random.randint(0, 1000000), :778
            st = rs.get_state()
            self.set_rng_state(st[1], st[2])
            os.makedirs('renders/env{}'.format(self.env_id), exist_ok=False)
            self._gif_frames = []

    def _gif_wanted(self):
        return True                                 # (hook: the Flat class saves only some episodes)

    def _gif_frame(self):
        a, b = np.asarray(self.obs_image, np.uint8), np.asarray(self.desired_goal, np.uint8)
        if a.shape != b.shape:
            return a.copy()
        return np.concatenate([a, np.zeros((a.shape[0], 4, 3), np.uint8), b], axis=1)

    def _gif_save(self):
        from PIL import Image
        tasknums = '-'.join(str(i) for i in np.where(self.desired_goal_vector[0] == 1)[0])
        completed = '-'.join(str(i) for i in np.where(self.achieved_goal_vector[0] == 1)[0])
        path = 'renders/env{}/E{}({})_{}({}).gif'.format(self.env_id, self.ep_no, self.step_num, tasknums, completed)
        imgs = [Image.fromarray(np.kron(f, np.ones((4, 4, 1), np.uint8))) for f in self._gif_frames]
        imgs[0].save(path, save_all=True, append_images=imgs[1:], duration=100, loop=0)
        return path

    # -- reference attributes derived from the device state on demand --------------------
    @property
    def obs_one_hot(self):
        if not self._has_reset:
            return None
        return self._vec.one_hot()[0].cpu().numpy().astype(int)

    @property
    def agent_pos(self):
        """ray.py:624-626: Coord(row, col, max_row=STATE_W - 1, max_col=STATE_H - 1) -- here a GridPos (coord.py: .row, .col, .tuple(),
        clamped + and -, and it still compares equal to / unpacks like the (row, col) pair this attribute used to be)."""
        if not self._has_reset:
            return None
        r, c = self._vec.agent_rc[0].cpu().tolist()
        return GridPos(r, c, self.STATE_W - 1, self.STATE_H - 1)

    @property
    def INIT_OBS_VECTOR(self):
        if not self._has_reset:
            return None
        st = self._vec.get_state()
        return _one_hot_from(st['init_grid'][0], st['init_agent_rc'][0], 0)

    @property
    def observation_vector(self):
        """ray.py:185-187, 354-356: the state-vector counterpart of `observation` -- obs_one_hot, the two goal vectors (the live (1, T) arrays
        that info carries) and INIT_OBS_VECTOR.  Built from the device state when read (None before the first reset(), ray.py:124)."""
        if not self._has_reset:
            return None
        return {'observation': self.obs_one_hot, 'desired_goal': self.desired_goal_vector, 'achieved_goal': self.achieved_goal_vector,
                'init_observation': self.INIT_OBS_VECTOR}

    @property
    def fixed_state_list(self):
        """ray.py:116-118: the fixed_init_state one-hot placements generate_fixed_states drew at construction (AttributeError when
        fixed_init_state == 0: the reference never sets the attribute then)."""
        if not self.fixed_init_state:
            raise AttributeError("%r object has no attribute 'fixed_state_list'" % type(self).__name__)
        S, out = self.STATE_W, []
        for cells in self._vec.fixed_states()[0]:
            oh = np.zeros((S, S, 12), dtype=int)
            for k in range(9):                      # channels 0-7 the objects, 8 the agent (ray.py:605-608)
                oh[cells[k] // S, cells[k] % S, k] = 1
            out.append(oh)
        return out

    def seed(self, seed=None):
        return self._vec.seed(seed)                                               # ray.py:145-147

    def generate_fixed_states(self, num_states=None):
        """ray.py:149-154: draw the fixed_init_state placements again from the env's CURRENT stream position (the constructor did it once,
        ray.py:116-118; a caller who pins the stream afterwards -- `env.np_random = RandomState(...)` -- redraws the pool here, as assigning
        `env.fixed_state_list = env.generate_fixed_states(k)` does in the reference).  The pool's size is fixed at construction."""
        if not self.fixed_init_state:
            raise ValueError('the env was built with fixed_init_state=0')
        if num_states is not None and int(num_states) != self.fixed_init_state:
            raise ValueError('the pool holds fixed_init_state=%d placements' % self.fixed_init_state)
        from . import _lib as L
        L.check(self._lib.cw_generate_fixed_states(self._eng, self._stream), 'cw_generate_fixed_states', self._lib)
        return self.fixed_state_list

    @property
    def np_random(self):
        """The env's RNG stream as a numpy RandomState (ray.py:146).  Reading gives a SNAPSHOT of the device-resident
        stream (drawing from it does not advance the env); assigning a RandomState -- `env.np_random =
        np.random.RandomState(12345)`, as reference users do to pin a stream -- injects its state."""
        k, p = self.get_rng_state()
        rs = np.random.RandomState()
        rs.set_state(('MT19937', k, p, 0, 0.0))
        return rs

    @np_random.setter
    def np_random(self, rs):
        st = rs.get_state()
        if st[0] != 'MT19937':
            raise ValueError('np_random must be a numpy RandomState (MT19937), as with gym <= 0.21')
        self.set_rng_state(st[1], st[2])

    def set_rng_state(self, key, pos):
        """Inject a numpy RandomState state (`env.np_random.set_state` in reference terms)."""
        self._vec.set_rng_states(np.asarray(key)[None], np.asarray([pos]))

    def get_rng_state(self):
        k, p = self._vec.get_rng_states()
        return k[0], int(p[0])

    def _pull_goals(self, force=False):
        ach, des = self._h_ach.item(0), self._h_des.item(0)
        if not force and ach == self._masks_seen[0] and des == self._masks_seen[1]:
            return                                  # (most steps change neither vector: nothing to rewrite)
        self._masks_seen = (ach, des)
        if self._bit_rows is not None:
            self.achieved_goal_vector[0, :] = self._bit_rows[ach]                 # mutated in place, like ray.py:659
            self.desired_goal_vector[0, :] = self._bit_rows[des]
        else:
            n = len(self.task_list)
            self.achieved_goal_vector[0, :] = [(ach >> i) & 1 for i in range(n)]
            self.desired_goal_vector[0, :] = [(des >> i) & 1 for i in range(n)]

    def _obs_dict(self):
        self.observation = {'observation': self.obs_image, 'desired_goal': self.desired_goal,
                            'achieved_goal': self.obs_image, 'init_observation': self.INIT_OBS}   # ray.py:194-196
        return self.observation

    def reset(self, render_next=False):
        if self.store_gif is True and self.step_num != 0 and self.ep_no % self.render_save_rate == 0 and self._gif_frames and self._gif_wanted():
            self._gif_save()                                                      # ray.py:160-167
        if self.step_num != 0:                                                    # ray.py:200-201
            self.ep_no += 1
        self.step_num = 0
        self._vec.reset()                                                          # returns after the stream sync
        self._has_reset = True
        if not self._live:
            self.obs_image[...] = self._h_obs
            self.desired_goal[...] = self._h_goal
            self.INIT_OBS[...] = self._h_init
        self._pull_goals(force=True)                # (the caller may have written into the vectors it was handed)
        if self.store_gif:                                                        # ray.py:205-216
            self._gif_frames = [self._gif_frame()] if self.ep_no % self.render_save_rate == 0 else []
        return self._obs_dict()

    def _after_step_enqueue(self):
        """hook: more work for the same stream sync (the one-hot façade exports its state here)"""

    def _exact_values(self):
        """hook (reference_dtypes=True only): values the engine's uint8 frame cannot hold, restored in the int64 copy"""

    def step(self, action):
        a = int(action)
        if not 0 <= a < 6:
            raise IndexError('list index out of range')                           # ACTIONS[action], ray.py:308
        if not self._has_reset:
            # the reference's own failure for a step() before the first reset(): step_num has been counted (ray.py:309), then agent_pos is None --
            # `None + Coord` for a move (ray.py:393), `None.tuple()` for pickup / drop (ray.py:315, :330)
            self.step_num += 1
            if a < 4:
                raise TypeError("unsupported operand type(s) for +: 'NoneType' and 'Coord'")
            raise AttributeError("'NoneType' object has no attribute 'tuple'")
        if self._resident:
            rc = self._step_resident(self._eng, a, self._want_onehot)               # doorbell + spin: no launch, no stream sync
        else:
            self._act[0] = a
            rc = self._lib.cw_step(self._eng, self._act_p, 0, self._stream)        # 0 = CW_ACT_I32
            self._after_step_enqueue()
            if rc == 0:
                rc = self._lib.cw_synchronize(self._eng, self._stream)
        if rc != 0:
            from . import _lib as L
            L.check(rc, 'cw_step')
        self.step_num += 1
        if not self._live:
            self.obs_image[...] = self._h_obs
            self._exact_values()
        self._pull_goals()
        if self.store_gif is True and self.ep_no % self.render_save_rate == 0:    # ray.py:370-374
            self._gif_frames.append(self._gif_frame())
        info = {'task_success': self.achieved_goal_vector, 'desired_goal': self.desired_goal_vector,
                'achieved_goal': self.achieved_goal_vector}                        # ray.py:376-378
        return self._obs_dict(), self._h_reward.item(0), self._h_done.item(0) != 0, info

    def render(self, state=None, mode='Non', tile_size=4):
        """render() of ray.py:442-520: the current state, or a caller-supplied (S,S,12) one-hot `state` of any content -- the image
        is the sum of the colours of the objects in each cell, the agent the first cell with channel 8 set, the held item's colour
        comes from the hold channels (cw_render_onehot).  With the default uint8 dtype sums above 255 wrap; reference_dtypes=True
        returns the reference's int64 image exactly."""
        if mode == 'human':
            raise NotImplementedError('mode="human" (matplotlib popup) is out of scope')
        if state is None:
            img = self._vec.render()[0].cpu().numpy().astype(self._dtype)
            return self._exact_image(img) if not self._live else img
        st = np.asarray(state)
        S = self.STATE_W
        if st.shape != (S, S, 12):
            raise ValueError('state must be a (%d, %d, 12) one-hot array' % (S, S))
        if not (st[:, :, 8] == 1).any():
            raise IndexError('index 0 is out of bounds for axis 0 with size 0')   # state_idxs[0][0], ray.py:454
        if st.min() < 0 or st.max() > 1:
            raise ValueError('state must be a one-hot (0/1) array')
        img = self._vec.render_states(st.astype(np.uint8)[None])[0].cpu().numpy()
        return img.astype(self._dtype)

    def _exact_image(self, img):
        return img

    def compute_reward(self, achieved_goal, desired_goal, info=None):
        return self._vec.compute_reward(achieved_goal, desired_goal, info)

    # -- the reference class's small public helpers (host convenience; the per-step values come from the kernels)
    def compute_reward_equal(self, achieved_goal=None, desired_goal=None, info=None):
        """ray.py:757-761: MAX_STEPS iff the two goal vectors are equal position by position, else -1 (whatever reward_style the env was built with)."""
        return self.MAX_STEPS if self.short_circuit_check(desired_goal, achieved_goal, 4) else -1

    def compute_reward_subset(self, achieved_goal=None, desired_goal=None, info=None):
        """ray.py:763-767: MAX_STEPS iff the largest element of desired - achieved is 0 (nothing desired is missing AND some position is equal), else -1."""
        d = np.asarray(desired_goal).astype(np.int64).reshape(-1)
        a = np.asarray(achieved_goal).astype(np.int64).reshape(-1)
        return self.MAX_STEPS if int((d - a).max()) == 0 else -1

    @staticmethod
    def short_circuit_check(a, b, n):
        """ray.py:747-755 compares a and b in n chunks and a tail: together they cover every position, so this is plain equality of the two vectors."""
        a, b = np.asarray(a).reshape(-1), np.asarray(b).reshape(-1)
        return a.shape == b.shape and bool((a == b).all())

    def one_hot(self, obj=None, agent=False, holding=None):
        """ray.py:784-792: one cell's 12-channel row as a list -- channel obj (0..7, OBJECTS order), channel 8 for the agent, channel 9 + holding."""
        row = [0] * 12
        if obj is not None:
            row[obj] = 1
        if agent:
            row[8] = 1
        if holding is not None:
            row[9 + holding] = 1
        return row

    @staticmethod
    def translate_one_hot(one_hot_row):
        """ray.py:794-799, the inverse: -> (object index or None, the agent channel's value, held item's index 0..2 or None)."""
        row = np.asarray(one_hot_row)
        objs, held = row[:8], row[9:]
        return (int(objs.argmax()) if objs.any() else None), row[8], (int(held.argmax()) if held.any() else None)

    def close(self):
        self._vec.close()

    # gym.Env's protocol beside step / reset / render / close / seed (the reference gets it from its base class)
    @property
    def unwrapped(self):
        return self

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __str__(self):
        return '<%s instance>' % type(self).__name__


class CraftingWorldEnvFlat(CraftingWorldEnv):
    """craftingworld_flat.py:46-198: Box observation (the frame only), 8x8 / 100 steps defaults,
    no fixed_init_state kwarg."""
    _default_size = (8, 8)
    _default_max_steps = 100

    def __init__(self, size=None, max_steps=None, store_gif=False, render_save_rate=1, task_list=TASK_LIST,
                 selected_tasks=TASK_LIST, number_of_tasks=None, stacking=True, reward_style=None, **kw):
        super().__init__(size=size, max_steps=max_steps, store_gif=store_gif, render_save_rate=render_save_rate,
                         task_list=task_list, selected_tasks=selected_tasks, number_of_tasks=number_of_tasks,
                         stacking=stacking, reward_style=reward_style, **kw)
        P = 4 * self.STATE_W
        self.observation_space = Box(low=0, high=255, shape=(P, P, 3), dtype=self._dtype)   # flat.py:57

    def _gif_wanted(self):
        # craftingworld_flat.py:64-71: the finished episode's GIF is written only if it achieved something or every 30th episode
        return bool((self.achieved_goal_vector[0] == 1).any()) or self.ep_no % 30 == 0

    def reset(self, render_next=False):
        super().reset()
        return self.obs_image                                                     # flat.py:119

    def step(self, action):
        _, r, d, info = super().step(action)
        return self.obs_image, r, d, info                                         # flat.py:185


def _one_hot_from(grid, agent_rc, hold):
    S = grid.shape[0]
    oh = np.zeros((S, S, 12), dtype=int)
    r, c = np.nonzero(grid)
    oh[r, c, grid[r, c] - 1] = 1
    oh[agent_rc[0], agent_rc[1], 8] = 1
    if hold:
        oh[agent_rc[0], agent_rc[1], 8 + hold] = 1
    return oh


class CraftingWorldEnvOneHot(CraftingWorldEnv):
    """carftingworld_onehot.py:53-389: observations are the (S,S,12) one-hot states instead of
    images (observation/achieved_goal = current, desired_goal = imagine_obs final state,
    init_observation = state at reset); dynamics identical."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        # resident steps leave obs_one_hot in the engine's pinned host buffer (cw_buffer_table.host_onehot); the launch path exports it
        # with a kernel of its own after every step
        self._resident = self._resident and self._vec._host_onehot is not None
        self._want_onehot = 1 if self._resident else 0
        S = self.STATE_W
        oh = lambda: Box(low=0, high=1, shape=(S, S, 12), dtype=self._dtype)  # noqa: E731
        self.observation_space = Dict(dict(observation=oh(), desired_goal=oh(), achieved_goal=oh(),
                                           init_observation=oh()))                # onehot.py:84-103 (dtype=int there; uint8 here unless reference_dtypes)
        self._oh_pin = torch.zeros((1, S, S, 12), dtype=torch.uint8).pin_memory()   # GPU-visible host memory (launch path, reset exports)
        self._oh_pin_np = self._oh_pin.numpy()[0]
        self._oh_pin_p = C.c_void_p(self._oh_pin.data_ptr())
        # where a step leaves obs_one_hot: the engine's pinned buffer (resident steps) or ours (the launch path's export kernel)
        self._oh_src = self._vec._host_onehot if self._resident else self._oh_pin_np
        self._oh_src_p = C.c_void_p(self._oh_src.ctypes.data)
        # default dtype uint8: `observation` IS that buffer, mutated in place by later steps like the frames of the other classes (and like
        # the reference's obs_one_hot, onehot.py:369-371); reference_dtypes=True: int64 copies
        self._oh = self._oh_src if self._live else np.zeros((S, S, 12), dtype=int)
        self._oh_goal = np.zeros((S, S, 12), dtype=self._dtype)
        self._oh_init = np.zeros((S, S, 12), dtype=self._dtype)

    def _after_step_enqueue(self):
        self._lib.cw_export_onehot(self._eng, self._oh_pin_p, self._stream)         # onehot.py:369-371

    def _oh_dict(self):
        self.observation = {'observation': self._oh, 'desired_goal': self._oh_goal, 'achieved_goal': self._oh,
                            'init_observation': self._oh_init}
        return self.observation

    def reset(self, render_next=False):
        super().reset()
        self._lib.cw_export_onehot_of(self._eng, 1, self._oh_pin_p, self._stream)   # goal state (onehot.py:310)
        self._lib.cw_synchronize(self._eng, self._stream)
        self._oh_goal[...] = self._oh_pin_np
        self._lib.cw_export_onehot_of(self._eng, 0, self._oh_src_p, self._stream)   # current state, into the buffer the steps rewrite
        self._lib.cw_synchronize(self._eng, self._stream)
        if not self._live:
            self._oh[...] = self._oh_src
        self._oh_init[...] = self._oh_src                                         # onehot.py:203
        return self._oh_dict()

    def step(self, action):
        _, r, d, info = super().step(action)
        if not self._live:
            self._oh[...] = self._oh_src
        return self._oh_dict(), r, d, info


class CraftingWorldEnvAltObs(CraftingWorldEnv):
    """craftingworld_altobs.py:85-886 (exported by the reference, not registered): same dynamics, 3x3-px CPV
    rasteriser with a "holding" strip, images ((W+1)*3, H*3, 3); stacked_obs=True returns the four images
    stacked (4, ., ., 3) instead of the Dict (altobs.py:116-119, 258-261, 408-412).  The reference's int image reaches
    2 x colour in ONE place -- sticks held over a sticks cell: pixel 0 of the agent's tile is (90, 164, 320), altobs.py:527-543.
    The engine's frames are uint8 and hold (90, 164, 64) there (a value no other state produces); with reference_dtypes=True the
    int64 copies handed out are exact: the one pixel is restored on the host.  (Default dtype uint8: modulo 256.)"""
    _raster = 'alt'

    def _exact_image(self, img):
        first = img[0:3 * self.STATE_W:3, 0::3]                # pixel 0 of every tile (a view): sticks items, count x (45, 82, 160)
        twice = (first[..., 0] == 90) & (first[..., 1] == 164) & (first[..., 2] == 64)
        if twice.any():
            first[..., 2][twice] = 320
        return img

    def _exact_values(self):
        self._exact_image(self.obs_image)

    def __init__(self, *a, stacked_obs=False, **kw):
        super().__init__(*a, **kw)
        self.stacked_obs = stacked_obs
        if stacked_obs is True:
            self.observation_space = Box(low=0, high=255, shape=(4,) + self._vec.frame_shape, dtype=self._dtype)

    def _stack(self, o):
        return np.stack([o['observation'], o['desired_goal'], o['achieved_goal'], o['init_observation']])

    def reset(self, render_next=False):
        o = super().reset()
        return self._stack(o) if self.stacked_obs is True else o

    def step(self, action):
        o, r, d, info = super().step(action)
        return (self._stack(o) if self.stacked_obs is True else o), r, d, info
