"""CPU tier: the registration path (reference: gym_craftingworld/__init__.py:5-18 -- three ids, each with kwargs {'stacking': True, 'render_save_rate': 10}).
Real gym / gymnasium are not in the image; tools/gym_stub is the stand-in registry the fixtures were captured under (register / registry / make with
gym <= 0.21's entry-point resolution).  In a subprocess with that package importable: importing gym_craftingworld_amd registers the same three ids, they
resolve to the HIP-backed classes with the reference's default kwargs, gym.make reaches the constructors -- which refuse to run without a GPU, loudly
-- and, with the oracle-backed fake engine in place of the HIP one (tests/fake_engine.py), builds envs that carry the kwargs and step."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import importlib, json, sys
sys.path[:0] = [%(stub)r, %(root)r, %(tests)r]
import gym                                   # the stand-in registry
assert 'gym_stub' in gym.__file__, gym.__file__
import gym_craftingworld_amd as cw           # registers at import (register_with_gym)
from gym.envs.registration import registry
out = {'ids': sorted(registry), 'resolved': {}, 'kwargs': {}, 'again': cw.register_with_gym()}
for env_id, (entry, kwargs) in registry.items():
    mod, cls = entry.split(':')
    out['resolved'][env_id] = getattr(importlib.import_module(mod), cls).__name__
    out['kwargs'][env_id] = kwargs
    assert getattr(importlib.import_module(mod), cls) is {'craftingworld-v3': cw.CraftingWorldEnv, 'craftingworldflat-v3': cw.CraftingWorldEnvFlat,
                                                          'craftingworldonehot-v3': cw.CraftingWorldEnvOneHot}[env_id]
assert cw.CraftingWorldEnvRay is cw.CraftingWorldEnv
from gym_craftingworld_amd.envs import CraftingWorldEnvRay, CraftingWorldEnvFlat, CraftingWorldEnvOneHot, CraftingWorldEnvAltObs   # the reference's import path
assert CraftingWorldEnvRay is cw.CraftingWorldEnv
try:                                         # no GPU here: the constructor is reached and refuses, loudly (no CPU fallback)
    gym.make('craftingworld-v3')
    out['no_gpu'] = 'constructed?!'
except Exception as exc:
    out['no_gpu'] = type(exc).__name__ + ': ' + str(exc)
import fake_engine
fake_engine.install(None)
made = {}
for env_id in sorted(registry):
    env = gym.make(env_id, size=(6, 6), max_steps=30, seed=3)
    o = env.reset()
    r = env.step(1)
    made[env_id] = dict(cls=type(env).__name__, stacking=env.stacking, render_save_rate=env.render_save_rate, size=env.STATE_W, max_steps=env.MAX_STEPS,
                        obs=('dict' if isinstance(o, dict) else list(o.shape)), reward=int(r[1]))
    env.close()
flat = gym.make('craftingworldflat-v3', seed=1)
made['flat_defaults'] = [flat.STATE_W, flat.MAX_STEPS]
flat.close()
# docs/source/envs/custom_envs.rst:5-16: a user's own id on the reference's entry-point path, and the documented loop (gen_info.rst:62-82)
from gym.envs.registration import register
register(id='craftingworldMyCustomEnv-v3', entry_point='gym_craftingworld_amd.envs:CraftingWorldEnvRay', kwargs={'stacking': True})
custom = gym.make('craftingworldMyCustomEnv-v3', size=(5, 5), max_steps=12, seed=2)
obs, done, n = custom.reset(), False, 0
while not done:
    obs, reward, done, _ = custom.step(custom.action_space.sample())
    n += 1
custom.reset()
made['custom'] = [type(custom).__name__, custom.stacking, custom.render_save_rate, n, custom.ep_no]
custom.close()
own = cw.make('craftingworldonehot-v3', size=(5, 5), render_save_rate=2, seed=1)      # the package's own make: the same table, no gym needed
made['own_make'] = [type(own).__name__, own.render_save_rate, own.stacking]
own.close()
out['made'] = made
print('RESULT ' + json.dumps(out))
'''


def test_reference_ids_resolve_to_the_hip_classes_with_the_reference_kwargs():
    script = SCRIPT % dict(stub=os.path.join(ROOT, 'tools', 'gym_stub'), root=ROOT, tests=os.path.join(ROOT, 'tests'))
    env = {k: v for k, v in os.environ.items() if k != 'PYTHONPATH'}
    p = subprocess.run([sys.executable, '-c', script], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith('RESULT ')][-1][7:])
    assert out['ids'] == ['craftingworld-v3', 'craftingworldflat-v3', 'craftingworldonehot-v3'] and out['again'] is True      # gym_craftingworld/__init__.py:5-18
    assert out['resolved'] == {'craftingworld-v3': 'CraftingWorldEnv', 'craftingworldflat-v3': 'CraftingWorldEnvFlat', 'craftingworldonehot-v3': 'CraftingWorldEnvOneHot'}
    assert all(kw == {'stacking': True, 'render_save_rate': 10} for kw in out['kwargs'].values()), out['kwargs']
    assert out['no_gpu'].startswith('CraftingWorldError') and 'no CPU fallback' in out['no_gpu'], out['no_gpu']
    m = out['made']
    for env_id, cls in out['resolved'].items():
        assert m[env_id]['cls'] == cls and m[env_id]['stacking'] is True and m[env_id]['render_save_rate'] == 10 and (m[env_id]['size'], m[env_id]['max_steps']) == (6, 30)
        assert m[env_id]['reward'] == -1
    assert m['craftingworld-v3']['obs'] == 'dict' and m['craftingworldonehot-v3']['obs'] == 'dict' and m['craftingworldflat-v3']['obs'] == [24, 24, 3]
    assert m['flat_defaults'] == [8, 100] and m['own_make'] == ['CraftingWorldEnvOneHot', 2, True]
    assert m['custom'][:3] == ['CraftingWorldEnv', True, 1] and 1 <= m['custom'][3] <= 12 and m['custom'][4] == 1      # (the documented loop ran to done; reset() counted the episode)
