#!/usr/bin/env python3
"""Capture golden vectors from the REFERENCE itself (build container only).

Runs /root/reference's CraftingWorldEnvRay / ...AltObs / ...Flat / ...OneHot (imported under tools/gym_stub) from injected
numpy RandomState states and writes small .npz fixtures to tests/golden/.  A fixture is data
only: config kwargs, the initial MT19937 state, the action sequence, and what the reference
returned / held after every step and every reset.  tests/test_oracle_golden.py replays them
through the C oracle; tests/test_hip_parity.py (gpu) replays them through the HIP engine.

    python tools/gen_golden.py            # regenerate every fixture
    python tools/gen_golden.py flat8_random onehot5_random   # only these
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from refharness import (bits, codes_from_onehot, crc, import_reference, make_ref_env,  # noqa: E402
                        scripted_action)

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden')

TASK_LIST = ['MakeBread', 'EatBread', 'BuildHouse', 'ChopTree', 'ChopRock', 'GoToHouse', 'MoveAxe',
             'MoveHammer', 'MoveSticks']

# name, env kwargs, rng seed, steps, policy ('random' | 'scripted'), full images kept for the first K resets
SCENARIOS = [
    ('ray21_random',      dict(size=(21, 21)),                                   12345, 1500, 'random',   2),
    ('ray21_scripted',    dict(size=(21, 21)),                                   777,   2500, 'scripted', 2),
    ('ray5_random',       dict(size=(5, 5), max_steps=50),                       4242,  6000, 'random',   3),
    ('ray5_subset',       dict(size=(5, 5), max_steps=40, reward_style='subset'), 99,   5000, 'random',   1),
    ('ray5_scripted',     dict(size=(5, 5), max_steps=60),                       31337, 4000, 'scripted', 1),
    ('ray8_scripted',     dict(size=(8, 8), max_steps=100),                      2024,  5000, 'scripted', 2),
    ('ray8_nostack',      dict(size=(8, 8), max_steps=100, stacking=False),      5,     3000, 'scripted', 1),
    ('ray8_selected',     dict(size=(8, 8), max_steps=80, number_of_tasks=2,
                               selected_tasks=['MoveAxe', 'EatBread', 'GoToHouse', 'ChopTree']), 6, 4000, 'scripted', 1),
    ('ray8_subset_sel',   dict(size=(8, 8), max_steps=80, reward_style='subset',
                               selected_tasks=['BuildHouse', 'MakeBread', 'MoveSticks']), 8, 4000, 'scripted', 1),
    ('ray6_fixedinit',    dict(size=(6, 6), max_steps=40, fixed_init_state=3),   11,    3000, 'random',   2),
    ('ray32_random',      dict(size=(32, 32)),                                   32,    700,  'random',   1),
    ('ray32_scripted',    dict(size=(32, 32), max_steps=300),                    3232,  1500, 'scripted', 1),
    ('ray4_tiny',         dict(size=(4, 4), max_steps=30),                       404,   3000, 'random',   1),
]
# CraftingWorldEnvAltObs (craftingworld_altobs.py): same dynamics, 3x3-px CPV rasteriser.  The reference's int image reaches 2 x colour
# when the agent holds sticks on a sticks cell ((45, 82, 160) -> (90, 164, 320)): CRCs are stored of the uint8 view (what the batch
# engine's uint8 frames hold) AND of the int16 view (exact; what the N=1 facade returns with reference_dtypes=True)
ALT_SCENARIOS = [
    ('alt21_scripted',    dict(size=(21, 21)),                                   2121,  1500, 'scripted', 2),
    ('alt5_random',       dict(size=(5, 5), max_steps=40),                       55,    4000, 'random',   2),
    ('alt8_scripted',     dict(size=(8, 8), max_steps=60, reward_style='subset'), 88,   3000, 'scripted', 1),
    ('alt4_double',       dict(size=(4, 4), max_steps=120),                      109,   5000, 'random',   1),   # sticks held over sticks on 9 steps: values up to 320
    # stacked_obs=True (altobs.py:116-119, 258-261, 408-412): reset() / step() return ONE array, the four images stacked; the fixture pins the
    # reference's stack (order, shape, dtype) by the CRC of what it returned
    ('alt6_stacked',      dict(size=(6, 6), max_steps=40, stacked_obs=True),     606,   3000, 'scripted', 1),
]


# The other two REGISTERED classes, captured through their own return conventions:
# CraftingWorldEnvFlat (craftingworld_flat.py: reset()/step() return the bare frame, 8x8 / 100-step defaults, no
# fixed_init_state kwarg) and CraftingWorldEnvOneHot (carftingworld_onehot.py: every observation is an (S,S,12) one-hot
# state, desired_goal = imagine_obs' final state un-rendered).  `{}` kwargs = the class's own defaults.
FLAT_SCENARIOS = [
    ('flat8_random',      dict(),                                                 808,   4000, 'random',   2),
    ('flat8_scripted',    dict(),                                                 818,   4000, 'scripted', 1),
    ('flat5_subset',      dict(size=(5, 5), max_steps=40, reward_style='subset'), 505,   4000, 'random',   1),
]
ONEHOT_SCENARIOS = [
    ('onehot21_scripted', dict(size=(21, 21)),                                    2112,  2000, 'scripted', 1),
    ('onehot5_random',    dict(size=(5, 5), max_steps=50),                        515,   5000, 'random',   2),
    ('onehot8_subset',    dict(size=(8, 8), max_steps=80, reward_style='subset',
                               selected_tasks=['BuildHouse', 'ChopTree', 'MoveSticks', 'EatBread']), 838, 4000, 'scripted', 1),
    ('onehot6_fixedinit', dict(size=(6, 6), max_steps=40, fixed_init_state=2),    626,   2500, 'random',   1),
]


def capture(cls, kwargs, seed, steps, policy, keep_images, env_name='CraftingWorldEnvRay'):
    rng = np.random.RandomState(seed)
    st = rng.get_state()
    key0, pos0 = st[1].copy(), int(st[2])
    env = make_ref_env(cls, rng, **kwargs)
    flat, onehot = env_name == 'CraftingWorldEnvFlat', env_name == 'CraftingWorldEnvOneHot'
    stacked, stack_crc, stack_shape = bool(kwargs.get('stacked_obs')), [], []

    def views(ret):
        """(observation, desired_goal, init_observation) as the class hands them out"""
        if flat:
            assert ret is env.obs_image
            return ret, env.desired_goal, env.INIT_OBS
        if stacked:                                   # np.stack([observation, desired_goal, achieved_goal, init_observation]), altobs.py:258-261
            assert isinstance(ret, np.ndarray) and ret.shape[0] == 4 and np.array_equal(ret[0], ret[2])
            stack_crc.append(crc(ret.astype(np.int16)))
            stack_shape[:] = list(ret.shape)
            return ret[0], ret[1], ret[3]
        assert ret['achieved_goal'] is ret['observation']
        return ret['observation'], ret['desired_goal'], ret['init_observation']
    pol_rng = np.random.RandomState(seed ^ 0x5EED)
    size = env.STATE_W

    R = dict(desired=[], grid=[], agent=[], rng_pos=[], rng_crc=[], obs_crc=[], desired_img_crc=[],
             init_img_crc=[], at_step=[], ep_no=[], goal_grid=[], goal_agent=[], obs_crc16=[], desired_img_crc16=[], init_img_crc16=[])
    S = dict(action=[], reward=[], done=[], achieved=[], agent=[], hold=[], step_num=[], grid_crc=[],
             obs_crc=[], grid=[], obs_crc16=[], obs_max=[])
    alt = env_name == 'CraftingWorldEnvAltObs'

    imgs_desired, imgs_obs = [], []

    def record_reset(t):
        o_obs, o_goal, o_init = views(env.reset())
        obs = {'observation': o_obs, 'desired_goal': o_goal, 'init_observation': o_init}
        codes, agent, hold = codes_from_onehot(env.obs_one_hot)
        assert hold == 0
        if onehot:                                    # the goal STATE itself is the observation here
            g_codes, g_agent, g_hold = codes_from_onehot(o_goal)
            assert g_hold == 0
            R['goal_grid'].append(g_codes)
            R['goal_agent'].append(g_agent)
        R['desired'].append(bits(env.desired_goal_vector))
        R['grid'].append(codes)
        R['agent'].append(agent)
        s = env.np_random.get_state()
        R['rng_pos'].append(int(s[2]))
        R['rng_crc'].append(crc(s[1].astype(np.uint32)))
        R['obs_crc'].append(crc(obs['observation'].astype(np.uint8)))
        R['desired_img_crc'].append(crc(obs['desired_goal'].astype(np.uint8)))
        R['init_img_crc'].append(crc(obs['init_observation'].astype(np.uint8)))
        if alt:
            R['obs_crc16'].append(crc(obs['observation'].astype(np.int16)))
            R['desired_img_crc16'].append(crc(obs['desired_goal'].astype(np.int16)))
            R['init_img_crc16'].append(crc(obs['init_observation'].astype(np.int16)))
        R['at_step'].append(t)
        R['ep_no'].append(env.ep_no)
        if len(imgs_desired) < keep_images:
            imgs_desired.append(obs['desired_goal'].astype(np.uint8))
            imgs_obs.append(obs['observation'].astype(np.uint8))

    record_reset(0)
    n_success = 0
    for t in range(steps):
        if policy == 'random' or pol_rng.rand() < 0.15:
            a = int(pol_rng.randint(6))
        else:
            a = scripted_action(env, pol_rng)
        ret, reward, done, info = env.step(a)
        obs = {'observation': views(ret)[0]}
        codes, agent, hold = codes_from_onehot(env.obs_one_hot)
        S['action'].append(a)
        S['reward'].append(int(reward))
        S['done'].append(bool(done))
        S['achieved'].append(bits(info['achieved_goal']))
        S['agent'].append(agent)
        S['hold'].append(hold)
        S['step_num'].append(env.step_num)
        S['grid_crc'].append(crc(codes))
        S['obs_crc'].append(crc(obs['observation'].astype(np.uint8)))
        if alt:
            S['obs_crc16'].append(crc(obs['observation'].astype(np.int16)))
            S['obs_max'].append(int(np.max(obs['observation'])))
        if size <= 8:
            S['grid'].append(codes)
        n_success += int(reward == env.MAX_STEPS)
        if done:
            record_reset(t + 1)

    kw = dict(kwargs)
    meta = dict(kwargs=kw, seed=seed, steps=steps, policy=policy, n_success=n_success, n_resets=len(R['desired']), env=env_name)
    if flat or onehot:            # what the ctor was given (maybe nothing: the class defaults) beside the effective values
        ck = dict(kwargs)
        if 'size' in ck:
            ck['size'] = list(ck['size'])
        meta['ctor_kwargs'] = ck
        kw.setdefault('size', (env.STATE_W, env.STATE_H))
        kw.setdefault('max_steps', env.MAX_STEPS)
    kw['size'] = list(kw['size'])
    out = dict(
        meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8),
        key0=key0.astype(np.uint32), pos0=np.int32(pos0),
        action=np.array(S['action'], np.int8), reward=np.array(S['reward'], np.int32),
        done=np.array(S['done'], np.uint8), achieved=np.array(S['achieved'], np.uint16),
        agent=np.array(S['agent'], np.uint8), hold=np.array(S['hold'], np.uint8),
        step_num=np.array(S['step_num'], np.int32), grid_crc=np.array(S['grid_crc'], np.uint32),
        obs_crc=np.array(S['obs_crc'], np.uint32),
        r_desired=np.array(R['desired'], np.uint16), r_grid=np.array(R['grid'], np.uint8),
        r_agent=np.array(R['agent'], np.uint8), r_rng_pos=np.array(R['rng_pos'], np.int32),
        r_rng_crc=np.array(R['rng_crc'], np.uint32), r_obs_crc=np.array(R['obs_crc'], np.uint32),
        r_desired_img_crc=np.array(R['desired_img_crc'], np.uint32),
        r_init_img_crc=np.array(R['init_img_crc'], np.uint32),
        r_at_step=np.array(R['at_step'], np.int32), r_ep_no=np.array(R['ep_no'], np.int32),
        img_desired=np.array(imgs_desired, np.uint8), img_obs=np.array(imgs_obs, np.uint8),
        final_obs=env.obs_image.astype(np.uint8),
    )
    if size <= 8:
        out['grid'] = np.array(S['grid'], np.uint8)
    if alt:
        out['obs_crc16'] = np.array(S['obs_crc16'], np.uint32)
        out['obs_max'] = np.array(S['obs_max'], np.int16)          # > 255 where sticks are held over sticks
        out['r_obs_crc16'] = np.array(R['obs_crc16'], np.uint32)
        out['r_desired_img_crc16'] = np.array(R['desired_img_crc16'], np.uint32)
        out['r_init_img_crc16'] = np.array(R['init_img_crc16'], np.uint32)
        out['final_obs16'] = np.asarray(env.obs_image).astype(np.int16)
    if stacked:                                       # CRC (int16 view) of every array reset() / step() returned, in call order, and its shape
        out['stack_crc16'] = np.array(stack_crc, np.uint32)
        out['stack_shape'] = np.array(stack_shape, np.int32)
    if onehot:
        out['r_goal_grid'] = np.array(R['goal_grid'], np.uint8)
        out['r_goal_agent'] = np.array(R['goal_agent'], np.uint8)
    return out, n_success


ALIAS_SCENARIOS = [
    # name, env kwargs, rng seed, policy seed: the op script of tests/golden_util.py (alias_script) run through the reference class -- what a caller sees
    # of the env's OBJECTS (np_random as the live generator, goal vectors rebound by reset(), negative action ids)
    ('ray5_alias', 'ray', dict(size=(5, 5), max_steps=30), 4711, 37),
    ('flat5_alias', 'flat', dict(size=(5, 5), max_steps=30), 4722, None),          # (the bare frame returned; policy seed searched for below)
    ('onehot5_alias', 'onehot', dict(size=(5, 5), max_steps=30), 4733, None),      # (observations are one-hot states; desired_goal the un-rendered goal state)
]


def capture_alias(cls, kwargs, seed, policy_seed):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
    from golden_util import A_CHECK_KEPT, A_KEEP, alias_script, run_alias_script
    rng = np.random.RandomState(seed)
    st = rng.get_state()
    env = make_ref_env(cls, rng, **kwargs)
    ops, args = alias_script()
    if policy_seed is None:                        # the first policy seed whose three kept terminal infos all have an achieved bit set
        for policy_seed in range(400):
            rng = np.random.RandomState(seed)
            env = make_ref_env(cls, rng, **kwargs)
            if (run_alias_script(env, ops, args, policy_seed)[ops == A_KEEP][:, 0] != 0).all():
                break
        rng = np.random.RandomState(seed)
        env = make_ref_env(cls, rng, **kwargs)
    rows = run_alias_script(env, ops, args, policy_seed)
    kept = rows[ops == A_KEEP]
    assert (kept[:, 0] != 0).all(), 'a kept terminal info with an achieved bit set: pick another seed'
    chk = rows[ops == A_CHECK_KEPT]
    assert (chk[:, 2:4] == 0).all() and np.array_equal(chk[:, :2], kept[:, :2]), 'the reference rebinds the goal vectors at reset (ray.py:170, 176)'
    kw = dict(kwargs)
    kw['size'] = list(kw['size'])
    meta = dict(kwargs=kw, seed=seed, policy_seed=policy_seed, env=cls.__name__, kind='alias')
    return dict(meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8), key0=st[1].astype(np.uint32), pos0=np.int32(st[2]),
                ops=ops, args=args, rows=rows)


def main():
    classes = import_reference()
    os.makedirs(OUT, exist_ok=True)
    for name, key, kwargs, seed, pseed in ALIAS_SCENARIOS:
        if sys.argv[1:] and name not in sys.argv[1:]:
            continue
        out = capture_alias(classes[key], kwargs, seed, pseed)
        path = os.path.join(OUT, name + '.npz')
        np.savez_compressed(path, **out)
        print('%-18s ops=%4d  %6.1f KB' % (name, len(out['ops']), os.path.getsize(path) / 1024))
    todo = ([(sc, 'ray', 'CraftingWorldEnvRay') for sc in SCENARIOS] + [(sc, 'altobs', 'CraftingWorldEnvAltObs') for sc in ALT_SCENARIOS] +
            [(sc, 'flat', 'CraftingWorldEnvFlat') for sc in FLAT_SCENARIOS] + [(sc, 'onehot', 'CraftingWorldEnvOneHot') for sc in ONEHOT_SCENARIOS])
    only = set(sys.argv[1:])
    todo = [t for t in todo if not only or t[0][0] in only]
    for (name, kwargs, seed, steps, policy, keep), key, env_name in todo:
        out, n_success = capture(classes[key], kwargs, seed, steps, policy, keep, env_name)
        path = os.path.join(OUT, name + '.npz')
        np.savez_compressed(path, **out)
        ach = np.bitwise_or.reduce(out['achieved']) if len(out['achieved']) else 0
        print('%-18s steps=%5d resets=%4d successes=%3d achieved-bits-seen=%s  %6.1f KB' % (
            name, steps, len(out['r_desired']), n_success, format(int(ach), '09b'), os.path.getsize(path) / 1024))


if __name__ == '__main__':
    main()
