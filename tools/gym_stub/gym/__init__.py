"""Minimal stand-in for the `gym` package (gym is not installed in this image, no network).

Used by tools/gen_golden.py and tools/diff_vs_reference.py, in the build container, to
import the read-only reference under /root/reference and capture golden vectors, and by
tests/test_registration.py (in a subprocess) as the registry the product's three ids are
registered in and resolved from.  It provides exactly the names the reference touches
 and no environment logic: GoalEnv, spaces.{Box,Dict,Discrete},
utils.seeding.np_random, envs.registration.register / registry, make (gym <= 0.21's entry-point
resolution).  Nothing in the product or bench.py imports this package.
"""
import importlib

from . import spaces  # noqa: F401
from .envs.registration import register, registry  # noqa: F401


class Env:
    metadata = {}
    observation_space = None
    action_space = None

    def seed(self, seed=None):
        return [seed]

    def close(self):
        pass


class GoalEnv(Env):
    pass


def make(env_id, **kwargs):
    entry_point, default_kwargs = registry[env_id]
    mod_name, cls_name = entry_point.split(':')
    cls = getattr(importlib.import_module(mod_name), cls_name)
    kw = dict(default_kwargs)
    kw.update(kwargs)
    return cls(**kw)
