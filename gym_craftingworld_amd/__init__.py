"""gym_craftingworld_amd -- MI355X-native batched CraftingWorld step/reset engine.

Drop-in for the hot path of lauradarcy/gym-craftingworld (CraftingWorldEnvRay.step()/reset(),
gym_craftingworld/envs/craftingworld_ray.py): the gym.Env surface for one env, a
gym.vector.VectorEnv-shaped surface for N envs, hand-written HIP kernels underneath
(csrc/, C ABI in include/craftingworld.h).  Importing the package needs neither a GPU nor the
built library; constructing an env needs both (there is no CPU fallback).
"""
from .vec_env import (ACTION_NAMES, COLORS, COLORS_H, COLORS_N, DOWN, DROP, LEFT, MAX_STEPS, OBJECTS, PICKUP, PICKUPABLE, RIGHT, STATE_H, STATE_W,  # noqa: F401
                      TASK_LIST, UP, CraftingWorldVecEnv)
from .env import CraftingWorldEnv, CraftingWorldEnvAltObs, CraftingWorldEnvFlat, CraftingWorldEnvOneHot  # noqa: F401
CraftingWorldEnvRay = CraftingWorldEnv          # the reference's name for it (envs/__init__.py:1)
from ._lib import CraftingWorldError  # noqa: F401
from .adapters import GymnasiumVecAdapter, MultiDeviceVecEnv  # noqa: F401

__version__ = '0.1.0'

# Same ids and default kwargs as the reference registration (gym_craftingworld/__init__.py:5-18),
# resolved to the HIP-backed classes, when a gym with a registry is importable (it is not in the
# build image; the classes are then used directly).
REGISTERED_IDS = {
    'craftingworld-v3': ('gym_craftingworld_amd:CraftingWorldEnv', {'stacking': True, 'render_save_rate': 10}),
    'craftingworldflat-v3': ('gym_craftingworld_amd:CraftingWorldEnvFlat', {'stacking': True, 'render_save_rate': 10}),
    'craftingworldonehot-v3': ('gym_craftingworld_amd:CraftingWorldEnvOneHot', {'stacking': True, 'render_save_rate': 10}),
}


def register_with_gym():
    try:
        from gym.envs.registration import register
    except Exception:  # noqa: BLE001
        return False
    for env_id, (entry, kwargs) in REGISTERED_IDS.items():
        try:
            register(id=env_id, entry_point=entry, kwargs=kwargs)
        except Exception:  # noqa: BLE001  (already registered)
            pass
    return True


def make(env_id, **kwargs):
    """gym.make for the three reference ids without needing gym."""
    import importlib
    entry, default = REGISTERED_IDS[env_id]
    mod, cls = entry.split(':')
    kw = dict(default)
    kw.update(kwargs)
    return getattr(importlib.import_module(mod), cls)(**kw)


register_with_gym()
