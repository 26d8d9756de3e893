"""Thin host-side adaptors over CraftingWorldVecEnv (no compute here).

* GymnasiumVecAdapter -- the gymnasium (>=0.26 / gymnasium.vector) calling convention:
  reset(seed=...) -> (obs, info), step() -> (obs, reward, terminated, truncated, info).
  terminated = the episode ended by success (reward == MAX_STEPS, ray.py:367), truncated = it ended
  by the step limit only.
* MultiDeviceVecEnv -- one engine + one stream per device driven from ONE process (SURVEY §8e's
  first option; bench.py uses the other, one process per GPU).  Envs are sharded contiguously over
  the devices (sharding.shard_range), no data-path collective; step() enqueues every shard's
  kernels before returning, so the devices run concurrently.
"""
import torch

from .sharding import shard_range
from .vec_env import CraftingWorldVecEnv


class GymnasiumVecAdapter:
    def __init__(self, venv):
        self.venv = venv
        self.num_envs = venv.num_envs
        self.single_observation_space = venv.single_observation_space
        self.single_action_space = venv.single_action_space
        self.observation_space = venv.observation_space
        self.action_space = venv.action_space

    def reset(self, *, seed=None, options=None):
        if seed is not None:
            self.venv.seed(seed)
        obs = self.venv.reset()
        return obs, {'desired_goal': self.venv.hdr[:, 6:8]}

    def step(self, actions):
        obs, reward, done, info = self.venv.step(actions)
        terminated = done & (reward == self.venv.MAX_STEPS)
        truncated = done & ~terminated
        info = dict(info, _episode=done)                 # gymnasium.vector: which rows of info['episode'] ({'r', 'l'}) hold a finished episode
        if 'terminal_observation' in info:
            info['final_observation'] = info['terminal_observation']
        return obs, reward, terminated, truncated, info

    def close(self):
        self.venv.close()


class MultiDeviceVecEnv:
    """num_envs envs over `devices` (e.g. ['cuda:0', ..., 'cuda:7']); env e's seed is seed+e whatever
    device owns it, so the concatenated result equals a single-device batch of num_envs envs."""

    def __init__(self, num_envs, devices, seed=0, env_menu=None, **kwargs):
        self.devices = [torch.device(d) for d in devices]
        self.num_envs = int(num_envs)
        self.ranges = [shard_range(g, len(self.devices), self.num_envs) for g in range(len(self.devices))]
        self.shards = []
        for (lo, hi), dev in zip(self.ranges, self.devices):
            kw = dict(kwargs)
            if env_menu is not None:
                kw['env_menu'] = env_menu[lo:hi]
            self.shards.append(CraftingWorldVecEnv(hi - lo, device=dev, seed=None if seed is None else seed + lo, **kw))
        self.streams = [torch.cuda.Stream(device=d) for d in self.devices]
        s0 = self.shards[0]
        self.single_observation_space, self.single_action_space = s0.single_observation_space, s0.single_action_space
        self.MAX_STEPS = s0.MAX_STEPS

    def set_rng_states(self, keys, pos):
        for (lo, hi), sh in zip(self.ranges, self.shards):
            sh.set_rng_states(keys[lo:hi], pos[lo:hi])

    def reset(self):
        out = []
        for sh, st in zip(self.shards, self.streams):
            with torch.cuda.stream(st):
                out.append(sh.reset())
        return out

    def step(self, actions):
        """actions: one tensor per device (already resident there), or a single host/device tensor of
        num_envs actions that is split.  Returns per-device lists [(obs, reward, done, info), ...];
        each shard's results are ordered on that shard's stream (self.streams[g])."""
        if torch.is_tensor(actions) or not isinstance(actions, (list, tuple)):
            actions = torch.as_tensor(actions)
            actions = [actions[lo:hi] for lo, hi in self.ranges]
        out = []
        for sh, st, a in zip(self.shards, self.streams, actions):
            with torch.cuda.stream(st):
                out.append(sh.step(a.to(sh.device, non_blocking=True)))
        return out

    def synchronize(self):
        for st in self.streams:
            st.synchronize()

    def close(self):
        for sh in self.shards:
            sh.close()
