#!/usr/bin/env python3
"""Live differential test (build container only): the REFERENCE itself vs the C oracle, step by
step, on many seeds and configs -- a wider net than the committed fixtures.  Compares reward, done,
achieved/desired vectors, one-hot state, observation / desired_goal / init_observation images and
the MT19937 state after every reset.  Prints a summary; exits non-zero on the first mismatch.

    python tools/diff_vs_reference.py [n_seeds [ray|alt|flat|onehot ...]]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from refharness import bits, codes_from_onehot, import_reference, make_ref_env, scripted_action  # noqa: E402
from oracle import OracleEnv  # noqa: E402

T = ['MakeBread', 'EatBread', 'BuildHouse', 'ChopTree', 'ChopRock', 'GoToHouse', 'MoveAxe', 'MoveHammer', 'MoveSticks']
CONFIGS = [
    (dict(size=(21, 21)), 400),
    (dict(size=(5, 5), max_steps=40), 600),
    (dict(size=(8, 8), max_steps=60, reward_style='subset'), 500),
    (dict(size=(6, 6), max_steps=30, stacking=False, selected_tasks=T[::-1]), 400),
    (dict(size=(7, 7), max_steps=35, fixed_init_state=5, number_of_tasks=3), 400),
    (dict(size=(4, 4), max_steps=20, selected_tasks=['MoveSticks', 'BuildHouse', 'GoToHouse', 'ChopTree', 'ChopRock']), 400),
    (dict(size=(32, 32), max_steps=80), 200),
    (dict(size=(12, 12), max_steps=100, reward_style='subset', selected_tasks=['EatBread', 'MakeBread'], number_of_tasks=1), 300),
]
# the same through CraftingWorldEnvAltObs (3x3-px CPV rasteriser); images compared as uint8 views
# CraftingWorldEnvFlat (bare frame returned; no fixed_init_state kwarg; {} = its own 8x8 / 100-step defaults) and
# CraftingWorldEnvOneHot (every observation a one-hot state; desired_goal = imagine_obs' final state, un-rendered)
FLAT_CONFIGS = [
    (dict(), 400),
    (dict(size=(5, 5), max_steps=30, reward_style='subset'), 400),
    (dict(size=(11, 11), max_steps=80, selected_tasks=T[2:7], number_of_tasks=3), 300),
]
ONEHOT_CONFIGS = [
    (dict(size=(21, 21), max_steps=150), 300),
    (dict(size=(5, 5), max_steps=30), 500),
    (dict(size=(7, 7), max_steps=40, fixed_init_state=3, reward_style='subset'), 400),
    (dict(size=(8, 8), max_steps=60, stacking=False, selected_tasks=T[::-1]), 300),
]
ALT_CONFIGS = [
    (dict(size=(21, 21), max_steps=120), 300),
    (dict(size=(5, 5), max_steps=30), 500),
    (dict(size=(9, 9), max_steps=60, fixed_init_state=2, reward_style='subset'), 400),
]


def one_hot(grid, agent, hold=0):
    oh = np.zeros(grid.shape + (12,), dtype=np.uint8)
    r, c = np.nonzero(grid)
    oh[r, c, grid[r, c] - 1] = 1
    oh[agent[0], agent[1], 8] = 1
    if hold:
        oh[agent[0], agent[1], 8 + hold] = 1
    return oh


def compare(env, ora, tag, variant='ray', returned=None):
    codes, agent, hold = codes_from_onehot(env.obs_one_hot)
    s = ora.state()
    assert np.array_equal(codes, s['grid']), (tag, 'grid')
    assert agent == s['agent'] and hold == s['hold'], (tag, 'agent/hold')
    assert bits(env.achieved_goal_vector) == s['achieved'] and bits(env.desired_goal_vector) == s['desired'], (tag, 'goals')
    if variant == 'onehot':        # obs_image IS obs_one_hot there (carftingworld_onehot.py:203)
        assert np.array_equal(env.obs_image.astype(np.uint8), one_hot(s['grid'], s['agent'], s['hold'])), (tag, 'one-hot observation')
        if returned is not None:
            assert np.array_equal(returned['observation'].astype(np.uint8), one_hot(s['grid'], s['agent'], s['hold'])), (tag, 'returned one-hot')
    else:
        assert np.array_equal(env.obs_image.astype(np.uint8), s['obs']), (tag, 'obs image')
        if variant == 'flat' and returned is not None:
            assert returned is env.obs_image, (tag, 'flat returns the frame itself')
    assert env.step_num == s['step_num'] and env.ep_no == s['ep_no'], (tag, 'counters')


def main(n_seeds):
    classes = import_reference()
    total_steps = total_resets = successes = 0
    t0 = time.time()
    todo = ([(classes['ray'], 'ray', kw, steps) for kw, steps in CONFIGS] + [(classes['altobs'], 'alt', kw, steps) for kw, steps in ALT_CONFIGS] +
            [(classes['flat'], 'flat', kw, steps) for kw, steps in FLAT_CONFIGS] + [(classes['onehot'], 'onehot', kw, steps) for kw, steps in ONEHOT_CONFIGS])
    only = set(sys.argv[2:])
    for ci, (cls, variant, kw, steps) in enumerate(todo):
        if only and variant not in only:
            continue
        alt = variant == 'alt'
        for seed in range(n_seeds):
            rng = np.random.RandomState(10_000 * ci + seed)
            st = rng.get_state()
            env = make_ref_env(cls, rng, **kw)
            okw = dict(kw)
            okw.setdefault('size', (env.STATE_W, env.STATE_H))       # (Flat's own defaults when none were given)
            okw.setdefault('max_steps', env.MAX_STEPS)
            ora = OracleEnv(rng_state=(st[1].copy(), int(st[2])), alt_obs=alt, **okw)
            pol = np.random.RandomState(seed)

            def do_reset():
                o = env.reset()
                oo = ora.reset()
                s = ora.state()
                if variant == 'onehot':
                    assert np.array_equal(o['desired_goal'].astype(np.uint8), one_hot(s['goal_grid'], s['goal_agent'])), (ci, seed, 'goal state')
                    assert np.array_equal(o['init_observation'].astype(np.uint8), one_hot(s['init_grid'], s['init_agent'])), (ci, seed, 'init state')
                else:
                    d_img, i_img = (env.desired_goal, env.INIT_OBS) if variant == 'flat' else (o['desired_goal'], o['init_observation'])
                    assert np.array_equal(d_img.astype(np.uint8), oo['desired_goal']), (ci, seed, 'desired_goal image')
                    assert np.array_equal(i_img.astype(np.uint8), oo['init_observation'])
                icodes, iagent, _ = codes_from_onehot(env.INIT_OBS_VECTOR)
                assert np.array_equal(icodes, s['init_grid'])
                k, p = ora.get_rng()
                rs = env.np_random.get_state()
                assert p == rs[2] and np.array_equal(k, rs[1]), (ci, seed, 'rng state')
                compare(env, ora, (ci, seed, 'reset'), variant, o)

            do_reset()
            total_resets += 1
            for t in range(steps):
                a = int(pol.randint(6)) if (seed % 2 == 0 or pol.rand() < 0.2) else scripted_action(env, pol)
                ret, r, d, info = env.step(a)
                _, r2, d2, info2 = ora.step(a)
                assert (r, d) == (r2, d2), (ci, seed, t, 'reward/done', r, r2, d, d2)
                compare(env, ora, (ci, seed, t), variant, ret)
                successes += int(r == env.MAX_STEPS)
                total_steps += 1
                if d:
                    do_reset()
                    total_resets += 1
    print('reference == oracle on %d steps, %d resets, %d successful episodes, %d configs x %d seeds (%.1f s)' % (
        total_steps, total_resets, successes, len([t for t in todo if not only or t[1] in only]), n_seeds, time.time() - t0))


def check_facade_helpers():
    """The N=1 facade's host helpers (compute_reward_equal / _subset, short_circuit_check, one_hot, translate_one_hot) against the reference's own
    methods (ray.py:747-767, 784-799).  They need no engine: the class is used without its constructor."""
    import itertools
    from gym_craftingworld_amd.env import CraftingWorldEnv as E
    ref = make_ref_env(import_reference()['ray'], np.random.RandomState(1), size=(5, 5), max_steps=40)
    mine = E.__new__(E)
    mine.MAX_STEPS = ref.MAX_STEPS
    rng = np.random.RandomState(0)
    vecs = [np.zeros(9, int), np.ones(9, int)] + [rng.randint(0, 2, 9) for _ in range(60)]
    for a, d in itertools.product(vecs, vecs):
        assert mine.compute_reward_equal(a, d) == ref.compute_reward_equal(a, d)
        assert mine.compute_reward_subset(a, d) == ref.compute_reward_subset(a, d)
        assert E.short_circuit_check(d, a, 4) == ref.short_circuit_check(d, a, 4)
    for obj, agent, holding in itertools.product([None] + list(range(8)), [False, True], [None, 0, 1, 2]):
        assert mine.one_hot(obj, agent, holding) == ref.one_hot(obj, agent, holding)
        row = np.array(ref.one_hot(obj, agent, holding))
        r1, r2 = E.translate_one_hot(row), type(ref).translate_one_hot(row)
        assert (r1[0], int(r1[1]), r1[2]) == (r2[0], int(r2[1]), r2[2]), (r1, r2)
    print('facade helpers == reference on %d goal-vector pairs and 72 one-hot rows' % (len(vecs) ** 2))


def check_facade_aliasing(n_seeds):
    """The N=1 facade's OBJECT behaviour against the reference's, op by op (tests/golden_util.py: run_alias_script): the committed script of
    tests/golden/ray5_alias.npz and randomised ones -- a kept terminal `info` across reset(), draws from / seed / set_state on env.np_random,
    an assigned RandomState, env.seed(), negative action ids -- on BOTH sides.  The facade runs on tests/fake_engine.py here (no GPU in the
    build container: the oracle steps; the GPU tier replays the fixture through the HIP engine), with both of its step paths and both dtypes."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
    import fake_engine
    import golden_util as G
    import gym_craftingworld_amd as cw
    refs = import_reference()
    variants = [('ray', cw.CraftingWorldEnv), ('flat', cw.CraftingWorldEnvFlat), ('onehot', cw.CraftingWorldEnvOneHot)]      # each against the reference class of its name

    configs = [dict(size=(5, 5), max_steps=30), dict(size=(6, 6), max_steps=20, reward_style='subset', fixed_init_state=3),
               dict(size=(8, 8), max_steps=40, stacking=False, selected_tasks=T[::-1])]
    n_ops = n_runs = 0
    for resident in (True, False):
        fake_engine.install(None, resident=resident)
        for ref_dt in (False, True):
            for seed in range(n_seeds + 1):
                key, mine_cls = variants[(seed // len(configs)) % len(variants)] if seed else variants[0]
                kw = dict(configs[seed % len(configs)])
                if key == 'flat':
                    kw.pop('fixed_init_state', None)    # (craftingworld_flat.py:52-55 has no such kwarg)
                ops, args = G.alias_script() if seed == 0 else G.random_alias_script(np.random.RandomState(900 + seed), 90)
                rng = np.random.RandomState(7000 + seed)
                st = rng.get_state()
                ref = make_ref_env(refs[key], rng, **kw)
                want = G.run_alias_script(ref, ops, args, seed)
                mine = mine_cls(reference_dtypes=ref_dt, **kw)
                mine.set_rng_state(st[1], int(st[2]))
                if kw.get('fixed_init_state'):
                    mine.generate_fixed_states()        # (the constructor drew the pool from its own seed: redraw it from the injected stream, as the reference's did)
                got = G.run_alias_script(mine, ops, args, seed)
                if not ref_dt:                          # uint8 frames are the engine's live buffers: a kept observation does not survive reset() (documented)
                    got[ops == G.A_CHECK_KEPT, G.ALIAS_KEPT_OBS_COL] = want[ops == G.A_CHECK_KEPT, G.ALIAS_KEPT_OBS_COL]
                bad = np.nonzero((got != want).any(axis=1))[0]
                assert bad.size == 0, (resident, ref_dt, seed, int(bad[0]), int(ops[bad[0]]), int(args[bad[0]]), got[bad[0]].tolist(), want[bad[0]].tolist())
                mine.close()
                n_ops += len(ops)
                n_runs += 1
    print('facade aliasing == reference on %d ops in %d scripts (committed + randomised; Ray / Flat / OneHot classes, resident / launch step paths, uint8 / int64 dtypes)' % (n_ops, n_runs))


if __name__ == '__main__':
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 12)
    check_facade_helpers()
    check_facade_aliasing(int(sys.argv[1]) if len(sys.argv) > 1 else 12)
