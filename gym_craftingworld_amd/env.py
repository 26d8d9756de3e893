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
import weakref

import numpy as np
import torch

from . import seeding
from .coord import GridPos
from .spaces import Box, Dict, Discrete
from .vec_env import ACTION_NAMES, TASK_LIST, CraftingWorldVecEnv


def _pinned_u8(shape):
    """GPU-visible (pinned) host memory a kernel can store into directly"""
    return torch.zeros(shape, dtype=torch.uint8).pin_memory()


class _EnvRandomState(np.random.RandomState):
    """`env.np_random` (ray.py:145-147): the env's generator ITSELF, as in the reference -- `env.np_random.seed(5)`, `.set_state(...)`, a
    `.randint(...)` or `.shuffle(...)` move the stream the env's next reset() draws from, and the draws of a reset() show in it.  The stream
    lives on the device (one MT19937 state per env, csrc/cw_mt.h); this object is its host mirror, kept coherent lazily: touching any public
    attribute first pulls the device's state if the engine has drawn since (reset(), generate_fixed_states()), and marks the mirror as
    possibly advanced -- the env pushes it back (cw_seed_mt) before the engine's next draw.  `get_state` only pulls."""
    _READ_ONLY = frozenset(('get_state',))

    def __init__(self, env):
        np.random.RandomState.__init__(self, 0)
        self._cw_env = weakref.ref(env)

    def __getattribute__(self, name):
        if name[0] != '_':
            ref = np.random.RandomState.__getattribute__(self, '_cw_env')
            env = ref() if ref is not None else None
            if env is not None:
                env._rng_host_access(self, name not in _EnvRandomState._READ_ONLY)
        return np.random.RandomState.__getattribute__(self, name)

    def __reduce__(self):                           # pickles / deep copies as a plain RandomState at the stream's current position (detached from the env)
        rs = np.random.RandomState()
        rs.set_state(self.get_state())
        return rs.__reduce__()


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
        # (what copy.deepcopy(env) builds its twin with: the reference env is plain Python and deep-copies -- planners do that --, this one holds an engine)
        self._ctor_kwargs = dict(size=size, fixed_init_state=fixed_init_state, max_steps=max_steps, store_gif=False, render_save_rate=render_save_rate,
                                 task_list=task_list, selected_tasks=selected_tasks, number_of_tasks=number_of_tasks, stacking=stacking,
                                 reward_style=reward_style, device=device, reference_dtypes=reference_dtypes, resident=resident)
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
        # np_random (ray.py:145-147): the host mirror of the device-resident stream (_EnvRandomState), or the caller's own RandomState once one was assigned
        self._np_random = _EnvRandomState(self)
        self._rng_stale, self._rng_dirty, self._rng_foreign, self._rng_foreign_seen = True, False, None, None
        # (the stream position the fixed_init_state pool was drawn from: a deep copy taken before the first reset() redraws it from there)
        self._pool_rng = seeding.mt_state_from_seed(v._seeds[0]) if fixed_init_state else None
        if not fixed_init_state:                    # (else the constructor has drawn the pool from the stream already: the mirror follows at its first use)
            key, pos = seeding.mt_state_from_seed(v._seeds[0])
            np.random.RandomState.set_state(self._np_random, ('MT19937', key, pos, 0, 0.0))   # the seeded state as numpy holds it, ray.py:70
            self._rng_stale = False
        # obs_one_hot (ray.py:119): kept current by every step from its first read on (the one-hot class: from construction)
        self._oh_track, self._oh_live, self._oh_src = False, None, None
        self._init_vec = None                       # INIT_OBS_VECTOR of the running episode, built on first read
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
            self.env_id = int(self.np_random.randint(0, 1000000))                  # np_random.randint(0, 1000000), :778: a draw from the env's own stream
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

    # -- step_num (ray.py:141, 203, 309): a plain int attribute of the reference that users may assign; the engine keeps its own copy in the env's
    # header (done = step_num >= MAX_STEPS is decided on the card), so an assignment after the first reset() is written through
    @property
    def step_num(self):
        return self._step_num

    @step_num.setter
    def step_num(self, value):
        value = int(value)
        if getattr(self, '_has_reset', False) and value != self._step_num:
            self._vec.set_state(step_num=np.array([value], np.int32))
        self._step_num = value

    # -- reference attributes derived from the device state on demand --------------------
    def _oh_setup(self):
        """obs_one_hot as a buffer every step keeps current: the resident stepper writes it into the engine's pinned host array
        (cw_buffer_table.host_onehot), the launch path exports it with a kernel of its own after every step (into a pinned array of ours)."""
        if self._oh_src is not None:
            return
        S = self.STATE_W
        self._oh_pin = _pinned_u8((1, S, S, 12))   # GPU-visible host memory (launch path, reset exports)
        self._oh_pin_np = self._oh_pin.numpy()[0]
        self._oh_pin_p = C.c_void_p(self._oh_pin.data_ptr())
        res = self._resident and self._vec._host_onehot is not None
        self._resident = res
        self._want_onehot = 1 if res else 0
        self._oh_src = self._vec._host_onehot if res else self._oh_pin_np
        self._oh_src_p = C.c_void_p(self._oh_src.ctypes.data)
        # default dtype uint8: obs_one_hot IS that buffer, mutated in place by later steps (like the reference's, ray.py:326-327); reference_dtypes=True:
        # an int64 array, a new one per episode (ray.py:179), rewritten after every step
        self._oh_live = self._oh_src if self._live else np.zeros((S, S, 12), dtype=int)
        self._oh_track = True

    def _oh_refresh(self, rebind=False):
        """the current state into the tracked buffer now (after reset() / a restore / the first read of obs_one_hot)"""
        self._lib.cw_export_onehot_of(self._eng, 0, self._oh_src_p, self._stream)
        self._lib.cw_synchronize(self._eng, self._stream)
        if not self._live:
            if rebind:
                self._oh_live = np.zeros(self._oh_src.shape, dtype=int)
            self._oh_live[...] = self._oh_src

    @property
    def obs_one_hot(self):
        """ray.py:119, 179: the (S,S,12) one-hot state.  The same array from read to read within an episode, kept current by every step from
        its first read on (uint8 view of the engine's pinned buffer; int64 with reference_dtypes=True).  Writing into it does NOT move the env: use set_state()."""
        if not self._has_reset:
            return None
        if not self._oh_track:
            self._oh_setup()
            self._oh_refresh()
        return self._oh_live

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
        """ray.py:183: the copy of obs_one_hot taken at reset(); one array per episode."""
        if not self._has_reset:
            return None
        if self._init_vec is None:
            st = self._vec.get_state()
            self._init_vec = _one_hot_from(st['init_grid'][0], st['init_agent_rc'][0], 0)
        return self._init_vec

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
        """ray.py:145-147: a NEW generator (the old np_random object is left behind, detached) seeded the gym <= 0.21 way; -> [seed]."""
        self._rng_detach()
        seeds = self._vec.seed(seed)
        self._np_random, self._rng_foreign = _EnvRandomState(self), None
        key, pos = seeding.mt_state_from_seed(seeds[0])     # what _vec.seed uploaded: the mirror shows the seeded state as numpy holds it ((key, 624); the
        np.random.RandomState.set_state(self._np_random, ('MT19937', key, pos, 0, 0.0))   # engine keeps the equivalent regenerated form)
        self._rng_stale, self._rng_dirty = False, False
        return seeds

    def generate_fixed_states(self, num_states=None):
        """ray.py:149-154: draw the fixed_init_state placements again from the env's CURRENT stream position (the constructor did it once,
        ray.py:116-118; a caller who pins the stream afterwards -- `env.np_random = RandomState(...)` -- redraws the pool here, as assigning
        `env.fixed_state_list = env.generate_fixed_states(k)` does in the reference).  The pool's size is fixed at construction."""
        if not self.fixed_init_state:
            raise ValueError('the env was built with fixed_init_state=0')
        if num_states is not None and int(num_states) != self.fixed_init_state:
            raise ValueError('the pool holds fixed_init_state=%d placements' % self.fixed_init_state)
        from . import _lib as L
        self._pool_rng = self.get_rng_state()        # (flushes what the host did to the generator first)
        L.check(self._lib.cw_generate_fixed_states(self._eng, self._stream), 'cw_generate_fixed_states', self._lib)
        self._rng_device_moved()
        return self.fixed_state_list

    # -- np_random: one stream, two homes (the device's MT19937 record and a numpy RandomState on the host) -------------------------------------
    @property
    def np_random(self):
        """The env's generator (ray.py:146), LIVE: what is drawn from it, or done to it (`seed`, `set_state`), moves the stream the next reset()
        draws from, and after a reset() it stands where the reset left the stream -- see _EnvRandomState.  Assigning a RandomState -- `env.np_random =
        np.random.RandomState(12345)`, as reference users do to pin a stream -- makes THAT object the env's generator: its state is read before every
        draw of the engine and written back after it."""
        return self._rng_foreign if self._rng_foreign is not None else self._np_random

    @np_random.setter
    def np_random(self, rs):
        if rs is self._np_random and self._rng_foreign is None:
            return
        if not isinstance(rs, np.random.RandomState) or rs.get_state()[0] != 'MT19937':
            raise ValueError('np_random must be a numpy RandomState (MT19937), as with gym <= 0.21')
        if isinstance(rs, _EnvRandomState):          # another env's live mirror: take its position, not the object
            plain = np.random.RandomState()
            plain.set_state(rs.get_state())
            rs = plain
        self._rng_detach()
        self._np_random, self._rng_foreign, self._rng_foreign_seen = _EnvRandomState(self), rs, None
        self._rng_stale, self._rng_dirty = True, False
        self._rng_flush()

    def _rng_detach(self):
        """the generator is about to be replaced (seed(), an assignment): the object callers may still hold stays behind as a plain generator standing
        where the env's stream stood (the reference's old RandomState does exactly that)"""
        if self._rng_foreign is None:
            self._rng_host_access(self._np_random, False)
            self._np_random._cw_env = None

    def _rng_host_access(self, mirror, may_draw):
        """_EnvRandomState's hook: the mirror is about to be read (may_draw False) or used"""
        if mirror is not self._np_random or self._rng_foreign is not None:
            return
        if self._rng_stale:
            self._rng_stale = False
            k, p = self._vec.get_rng_states()
            st = np.random.RandomState.get_state(mirror)
            np.random.RandomState.set_state(mirror, ('MT19937', k[0], int(p[0]), st[3], st[4]))   # (a cached gaussian stays the host object's own)
        if may_draw:
            self._rng_dirty = True

    def _rng_flush(self):
        """before the engine draws (reset, generate_fixed_states) or its stream is read: what the host did to the generator goes to the device"""
        if self._rng_foreign is not None:
            st = self._rng_foreign.get_state()       # (a plain RandomState cannot be hooked: look at it; upload only if the caller has moved it since our last look)
            seen = self._rng_foreign_seen
            if seen is None or st[2] != seen[1] or not np.array_equal(st[1], seen[0]):
                self._vec.set_rng_states(np.asarray(st[1])[None], np.asarray([st[2]]))
                self._rng_foreign_seen = (np.array(st[1], np.uint32), int(st[2]))
        elif self._rng_dirty:
            self._rng_dirty = False
            st = np.random.RandomState.get_state(self._np_random)
            self._vec.set_rng_states(np.asarray(st[1])[None], np.asarray([st[2]]))

    def _rng_device_moved(self):
        """after the engine drew: the host object follows (at once for a caller's own RandomState, at its next use for the mirror)"""
        if self._rng_foreign is not None:
            k, p = self._vec.get_rng_states()
            st = self._rng_foreign.get_state()
            self._rng_foreign.set_state(('MT19937', k[0], int(p[0]), st[3], st[4]))
            self._rng_foreign_seen = (k[0].copy(), int(p[0]))
        else:
            self._rng_stale = True

    def set_rng_state(self, key, pos):
        """Inject a numpy RandomState state (`env.np_random.set_state` in reference terms)."""
        self._vec.set_rng_states(np.asarray(key)[None], np.asarray([pos]))
        np.random.RandomState.set_state(self.np_random, ('MT19937', np.asarray(key, np.uint32), int(pos), 0, 0.0))   # (the host object shows it verbatim)
        self._rng_stale, self._rng_dirty = False, False
        self._rng_foreign_seen = (np.array(key, np.uint32), int(pos)) if self._rng_foreign is not None else None

    def get_rng_state(self):
        self._rng_flush()
        k, p = self._vec.get_rng_states()
        return k[0], int(p[0])

    def _bits(self, mask):
        if self._bit_rows is not None:
            return self._bit_rows[mask]
        return [(mask >> i) & 1 for i in range(len(self.task_list))]

    def _pull_goals(self, rebind=False):
        ach, des = self._h_ach.item(0), self._h_des.item(0)
        if rebind:                                  # reset(): two NEW arrays (ray.py:170, 176) -- the info of the finished episode keeps its vectors
            self.desired_goal_vector = np.zeros((1, len(self.task_list)), dtype=int)
            self.achieved_goal_vector = np.zeros((1, len(self.task_list)), dtype=int)
        elif ach == self._masks_seen[0] and des == self._masks_seen[1]:
            return                                  # (most steps change neither vector: nothing to rewrite)
        self._masks_seen = (ach, des)
        self.achieved_goal_vector[0, :] = self._bits(ach)                         # mutated in place within an episode, like ray.py:659
        self.desired_goal_vector[0, :] = self._bits(des)

    def _obs_dict(self):
        self.observation = {'observation': self.obs_image, 'desired_goal': self.desired_goal,
                            'achieved_goal': self.obs_image, 'init_observation': self.INIT_OBS}   # ray.py:194-196
        return self.observation

    def _pull_frames(self, rebind=False):
        """reference_dtypes=True: the int64 copies the caller holds.  reset() makes NEW arrays (ray.py:191-193: a kept observation of the finished episode
        survives, as in the reference); steps rewrite obs_image in place (ray.py:550-557).  Default dtype: the arrays ARE the engine's buffers."""
        if self._live:
            return
        if rebind:
            self.obs_image = self._h_obs.astype(self._dtype)
            self.desired_goal = self._h_goal.astype(self._dtype)
            self.INIT_OBS = self._h_init.astype(self._dtype)
        else:
            self.obs_image[...] = self._h_obs
        self._exact_values()

    def reset(self, render_next=False):
        if self.store_gif is True and self.step_num != 0 and self.ep_no % self.render_save_rate == 0 and self._gif_frames and self._gif_wanted():
            self._gif_save()                                                      # ray.py:160-167
        if self._step_num != 0:                                                   # ray.py:200-201
            self.ep_no += 1
        self._step_num = 0
        self._rng_flush()                           # (what the caller drew from / did to np_random since the engine's last draw)
        self._vec.reset()                                                          # returns after the stream sync
        self._rng_device_moved()
        self._has_reset = True
        self._init_vec = None
        self._pull_frames(rebind=True)
        self._pull_goals(rebind=True)
        if self._oh_track:
            self._oh_refresh(rebind=True)
        if self.store_gif:                                                        # ray.py:205-216
            self._gif_frames = [self._gif_frame()] if self.ep_no % self.render_save_rate == 0 else []
        return self._obs_dict()

    def _after_step_enqueue(self):
        """the launch path: more work for the same stream sync -- the tracked obs_one_hot is exported by a kernel of its own"""
        if self._oh_track:
            self._lib.cw_export_onehot(self._eng, self._oh_pin_p, self._stream)     # ray.py:326-327 / onehot.py:369-371

    def _exact_values(self):
        """hook (reference_dtypes=True only): values the engine's uint8 frame cannot hold, restored in the int64 copy"""

    def step(self, action):
        a = int(action)
        if not -6 <= a < 6:
            raise IndexError('list index out of range')                           # ACTIONS[action], ray.py:308
        if a < 0:
            a += 6                                  # a Python list index: ACTIONS[-1] is 'drop', ACTIONS[-6] 'up'
        if not self._has_reset:
            # the reference's own failure for a step() before the first reset(): step_num has been counted (ray.py:309), then agent_pos is None --
            # `None + Coord` for a move (ray.py:393), `None.tuple()` for pickup / drop (ray.py:315, :330)
            self._step_num += 1
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
        self._step_num += 1
        if not self._live:
            self._pull_frames()
            if self._oh_track:
                self._oh_live[...] = self._oh_src
        self._pull_goals()
        if self.store_gif is True and self.ep_no % self.render_save_rate == 0:    # ray.py:370-374
            self._gif_frames.append(self._gif_frame())
        info = {'task_success': self.achieved_goal_vector, 'desired_goal': self.desired_goal_vector,
                'achieved_goal': self.achieved_goal_vector}                        # ray.py:376-378
        return self._obs_dict(), self._h_reward.item(0), self._h_done.item(0) != 0, info

    # -- the env's whole state as data (SURVEY 5 "checkpoint / resume"; the reference's de-facto state is the attribute set of ray.py:119-141) ------
    _STATE_NDIM = dict(grid=3, init_grid=3, goal_grid=3, agent_rc=2, init_agent_rc=2, goal_agent_rc=2, hold=1, achieved=1, desired=1, step_num=1, ep_no=1)

    def _sync_counters_down(self):
        """ep_no is a plain attribute here (GIF names, render_save_rate): the engine's copy follows before the state is read"""
        self._vec.set_state(ep_no=np.array([self.ep_no], np.int32))

    def get_state(self):
        """-> dict of numpy arrays: CraftingWorldVecEnv.get_state()'s fields for this one env (leading dimension 1: grid, init_grid, goal_grid, agent_rc,
        init_agent_rc, goal_agent_rc, hold, achieved, desired, step_num, ep_no) + rng_key [1,624] / rng_pos [1] (np_random.get_state()[1:3]).
        `other.set_state(**env.get_state())` makes `other` continue exactly like `env`."""
        if not self._has_reset:
            raise RuntimeError('get_state() before the first reset(): there is no state yet (ray.py:119-124)')
        self._rng_flush()
        self._sync_counters_down()
        st = self._vec.get_state()
        st['rng_key'], st['rng_pos'] = self._vec.get_rng_states()
        return st

    def set_state(self, **fields):
        """Overwrite any subset of get_state()'s fields (arrays with or without the leading dimension of 1).  The frames, goal vectors, obs_one_hot,
        step_num and ep_no this object shows follow."""
        if not self._has_reset:
            raise RuntimeError('set_state() before the first reset()')
        rk, rp = fields.pop('rng_key', None), fields.pop('rng_pos', None)
        if (rk is None) != (rp is None):
            raise ValueError('rng_key and rng_pos go together')
        for k in fields:
            if k not in self._STATE_NDIM:
                raise ValueError('cannot set %r' % k)
        batched = {k: (np.asarray(a)[None] if np.ndim(a) == self._STATE_NDIM[k] - 1 else np.asarray(a)) for k, a in fields.items()}
        if batched:
            self._vec.set_state(**batched)
        if rk is not None:
            self.set_rng_state(np.asarray(rk).reshape(-1), int(np.asarray(rp).reshape(-1)[0]))
        self._after_restore()

    def save_checkpoint(self, path):
        """One file holding everything this env needs to continue bit-identically (cw_checkpoint_save: state, episode records, RNG stream, pool, counters)."""
        if not self._has_reset:
            raise RuntimeError('save_checkpoint() before the first reset()')
        self._rng_flush()
        self._sync_counters_down()
        self._vec.save_checkpoint(path)

    def load_checkpoint(self, path):
        """Resume from a save_checkpoint() file of an env built with the same configuration (ValueError otherwise); no reset() needed first."""
        self._vec.load_checkpoint(path)
        self._has_reset = True
        self._rng_dirty = False
        self._rng_device_moved()
        self._after_restore()

    def __deepcopy__(self, memo):
        """copy.deepcopy(env) -- what a planner does with the reference's plain-Python env to try actions on a copy: a second env on an engine of its own, in
        the same state down to the RNG stream, the fixed_init_state pool, the episode's frames and counters (an in-memory checkpoint round trip), and from
        then on independent of this one.  GIF recording is not copied."""
        twin = type(self)(**self._ctor_kwargs)
        if self._has_reset:
            from . import _lib as L
            self._rng_flush()
            self._sync_counters_down()
            n = int(self._lib.cw_checkpoint_bytes(self._eng))
            buf = np.empty(n, dtype=np.uint8)
            L.check(self._lib.cw_checkpoint_save(self._eng, buf.ctypes.data_as(C.c_void_p), n), 'cw_checkpoint_save', self._lib)
            L.check(twin._lib.cw_checkpoint_load(twin._eng, buf.ctypes.data_as(C.c_void_p), n), 'cw_checkpoint_load', twin._lib)
            twin._vec._has_reset = twin._has_reset = True
            twin._rng_dirty = False
            twin._rng_device_moved()
            twin._after_restore()
        else:
            if self._pool_rng is not None:           # (no state to checkpoint yet: the pool is redrawn from where this env drew it)
                twin.set_rng_state(*self._pool_rng)
                twin.generate_fixed_states()
            twin.set_rng_state(*self.get_rng_state())
            twin._step_num, twin.ep_no = self._step_num, self.ep_no
        twin._pool_rng = self._pool_rng
        memo[id(self)] = twin
        return twin

    def _after_restore(self):
        st = self._vec.get_state()
        self._step_num, self.ep_no = int(st['step_num'][0]), int(st['ep_no'][0])
        self._init_vec = None
        ach, des = int(st['achieved'][0]), int(st['desired'][0])
        self.achieved_goal_vector[0, :] = self._bits(ach)
        self.desired_goal_vector[0, :] = self._bits(des)
        self._masks_seen = (ach, des)
        self._h_ach[0], self._h_des[0] = ach, des    # (the step outputs the vectors are compared with: they are rewritten by the next step)
        self._pull_frames(rebind=True)
        if self._oh_track:
            self._oh_refresh(rebind=True)
        self._after_restore_extra()

    def _after_restore_extra(self):
        """hook: what else a subclass shows of the restored state"""


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
    """craftingworld_flat.py:46-198: Box observation (the frame only), 8x8 / 100 steps defaults, and NO fixed_init_state kwarg: passing one is
    a TypeError, as in the reference (craftingworld_flat.py:52-55).  **kw takes this package's own extras only (device, reference_dtypes, seed,
    resident)."""
    _default_size = (8, 8)
    _default_max_steps = 100

    def __init__(self, size=None, max_steps=None, store_gif=False, render_save_rate=1, task_list=TASK_LIST,
                 selected_tasks=TASK_LIST, number_of_tasks=None, stacking=True, reward_style=None, **kw):
        for k in kw:
            if k not in ('device', 'reference_dtypes', 'seed', 'resident'):
                raise TypeError("__init__() got an unexpected keyword argument %r" % k)
        super().__init__(size=size, max_steps=max_steps, store_gif=store_gif, render_save_rate=render_save_rate,
                         task_list=task_list, selected_tasks=selected_tasks, number_of_tasks=number_of_tasks,
                         stacking=stacking, reward_style=reward_style, **kw)
        self._ctor_kwargs.pop('fixed_init_state', None)
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
        # obs_one_hot is tracked from the start (_oh_setup): resident steps leave it in the engine's pinned host buffer (cw_buffer_table.host_onehot),
        # the launch path exports it with a kernel of its own after every step.  Default dtype uint8: `observation` IS that buffer, mutated in place by
        # later steps like the frames of the other classes (and like the reference's obs_one_hot, onehot.py:369-371); reference_dtypes=True: int64 copies
        self._oh_setup()
        S = self.STATE_W
        oh = lambda: Box(low=0, high=1, shape=(S, S, 12), dtype=self._dtype)  # noqa: E731
        self.observation_space = Dict(dict(observation=oh(), desired_goal=oh(), achieved_goal=oh(),
                                           init_observation=oh()))                # onehot.py:84-103 (dtype=int there; uint8 here unless reference_dtypes)
        self._oh_goal = np.zeros((S, S, 12), dtype=self._dtype)
        self._oh_init = np.zeros((S, S, 12), dtype=self._dtype)
        self._oh_scratch = _pinned_u8((1, S, S, 12))   # (not the tracked buffer: that one is the live observation)
        self._oh_scratch_np, self._oh_scratch_p = self._oh_scratch.numpy()[0], C.c_void_p(self._oh_scratch.data_ptr())

    def _oh_dict(self):
        self.observation = {'observation': self._oh_live, 'desired_goal': self._oh_goal, 'achieved_goal': self._oh_live,
                            'init_observation': self._oh_init}
        return self.observation

    def _pull_goal_and_init_states(self):
        S = self.STATE_W
        self._oh_goal, self._oh_init = np.zeros((S, S, 12), dtype=self._dtype), np.zeros((S, S, 12), dtype=self._dtype)   # new per episode (onehot.py:203, :310)
        for which, dst in ((1, self._oh_goal), (2, self._oh_init)):               # goal state (onehot.py:310), state at reset (:203)
            self._lib.cw_export_onehot_of(self._eng, which, self._oh_scratch_p, self._stream)
            self._lib.cw_synchronize(self._eng, self._stream)
            dst[...] = self._oh_scratch_np

    def reset(self, render_next=False):
        super().reset()                             # (the tracked current state is refreshed there)
        self._pull_goal_and_init_states()
        return self._oh_dict()

    def _after_restore_extra(self):
        self._pull_goal_and_init_states()

    def step(self, action):
        _, r, d, info = super().step(action)
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
        self._ctor_kwargs['stacked_obs'] = stacked_obs
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
