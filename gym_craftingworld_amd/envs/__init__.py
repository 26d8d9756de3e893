"""The reference's import path for its classes (gym_craftingworld/envs/__init__.py:1-4): `from gym_craftingworld.envs import CraftingWorldEnvRay`
becomes `from gym_craftingworld_amd.envs import CraftingWorldEnvRay` -- the same four names, bound to the HIP-backed N=1 classes of ../env.py."""
from ..env import CraftingWorldEnv as CraftingWorldEnvRay  # noqa: F401
from ..env import CraftingWorldEnvAltObs, CraftingWorldEnvFlat, CraftingWorldEnvOneHot  # noqa: F401
