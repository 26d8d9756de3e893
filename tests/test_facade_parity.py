"""GPU tier: the N=1 facade classes on the HIP engine against what the REFERENCE class did, object for object (tests/golden/ray5_alias.npz,
captured by tools/gen_golden.py from CraftingWorldEnvRay itself): np_random as the live generator (ray.py:145-147), goal vectors rebound by
reset() so that a kept terminal `info` survives (ray.py:170, 176), negative action ids (ACTIONS[action], ray.py:308); the same for randomised
op scripts against the facade on the oracle-backed fake engine (which tools/diff_vs_reference.py holds against the live reference); and the
public state access of the N=1 classes (get_state / set_state / save_checkpoint / load_checkpoint).  Bit-exact."""
import numpy as np
import pytest

import fake_engine
from golden_util import (A_CHECK_KEPT, ALIAS_KEPT_OBS_COL, alias_script, load, random_alias_script, run_alias_script)

pytestmark = pytest.mark.gpu

TASKS = ['MakeBread', 'EatBread', 'BuildHouse', 'ChopTree', 'ChopRock', 'GoToHouse', 'MoveAxe', 'MoveHammer', 'MoveSticks']


def _explain(ops, args, got, want):
    bad = np.nonzero((got != want).any(axis=1))[0]
    return 'op #%d (code %d, arg %d): got %s, expected %s' % (bad[0], ops[bad[0]], args[bad[0]], got[bad[0]].tolist(), want[bad[0]].tolist()) if bad.size else ''


@pytest.mark.parametrize('reference_dtypes', [False, True])
@pytest.mark.parametrize('resident', [True, False])
@pytest.mark.parametrize('name', ['ray5_alias', 'flat5_alias', 'onehot5_alias'])
def test_alias_fixture_replays_through_the_hip_engine(name, resident, reference_dtypes):
    """each fixture was captured from the reference class of that name (Ray: a dict of images; Flat: the bare frame; OneHot: a dict of one-hot states)"""
    import gym_craftingworld_amd as cw
    meta, kw, g = load(name)
    ops, args = alias_script()
    assert np.array_equal(ops, g['ops']) and np.array_equal(args, g['args'])
    env = getattr(cw, meta['env'])(reference_dtypes=reference_dtypes, resident=resident, **kw)
    assert env._resident == resident
    env.set_rng_state(g['key0'], int(g['pos0']))
    rows = run_alias_script(env, ops, args, meta['policy_seed'])
    want = g['rows'].copy()
    if not reference_dtypes:          # the default uint8 frames are the engine's live buffers (craftingworld.h: valid until the next reset): a kept frame shows the new episode
        keep = ops == A_CHECK_KEPT
        assert (rows[keep, ALIAS_KEPT_OBS_COL] != want[keep, ALIAS_KEPT_OBS_COL]).any()
        rows[keep, ALIAS_KEPT_OBS_COL] = want[keep, ALIAS_KEPT_OBS_COL]
    assert np.array_equal(rows, want), _explain(ops, args, rows, want)
    env.close()


@pytest.mark.parametrize('case', [dict(size=(5, 5), max_steps=30), dict(size=(6, 6), max_steps=20, reward_style='subset', fixed_init_state=3),
                                  dict(size=(8, 8), max_steps=40, stacking=False, selected_tasks=TASKS[::-1]), dict(size=(21, 21), max_steps=25)],
                         ids=['5x5', '6x6_pool_subset', '8x8_nostack', '21x21'])
def test_random_alias_scripts_engine_vs_oracle_backed_facade(case, monkeypatch):
    """randomised op scripts (draws from / seed / set_state on np_random, assigned generators, env.seed(), kept infos, negative ids) through the
    facade on the HIP engine and through the SAME facade code on tests/fake_engine.py, where the oracle steps: every row equal"""
    import gym_craftingworld_amd as cw
    import gym_craftingworld_amd.env as E
    real_vec = E.CraftingWorldVecEnv
    for seed in range(6):
        ops, args = random_alias_script(np.random.RandomState(4000 + seed), 120)
        st = np.random.RandomState(5000 + seed).get_state()
        rows = []
        for side in ('hip', 'oracle'):
            with monkeypatch.context() as m:
                if side == 'oracle':
                    fake_engine.install(m, resident=bool(seed % 2))
                else:
                    assert E.CraftingWorldVecEnv is real_vec
                env = cw.CraftingWorldEnv(reference_dtypes=bool(seed & 2), resident=bool(seed % 2), **case)
                env.set_rng_state(st[1], int(st[2]))
                if case.get('fixed_init_state'):
                    env.generate_fixed_states()
                rows.append(run_alias_script(env, ops, args, seed))
                env.close()
        assert np.array_equal(rows[0], rows[1]), (seed, _explain(ops, args, rows[0], rows[1]))


@pytest.mark.parametrize('cls_name', ['CraftingWorldEnv', 'CraftingWorldEnvOneHot', 'CraftingWorldEnvFlat', 'CraftingWorldEnvAltObs'])
def test_single_env_state_access_and_checkpoints(cls_name, tmp_path):
    """get_state() -> set_state() into a second env and save_checkpoint() -> load_checkpoint() into a third (no reset() first): all three continue
    identically -- observations of every kind, rewards, goal vectors, counters, the RNG stream across the following resets."""
    import gym_craftingworld_amd as cw
    cls = getattr(cw, cls_name)
    kw = dict(size=(7, 7), max_steps=35, reward_style='subset')
    a, b, c = cls(seed=5, **kw), cls(seed=6, **kw), cls(seed=7, **kw)

    def frame(o):
        return o if isinstance(o, np.ndarray) else o['observation']
    a.reset(), b.reset()
    pol = np.random.RandomState(12)
    for _ in range(17):
        if a.step(int(pol.randint(-6, 6)))[2]:
            a.reset()
    a.ep_no = 41
    path = str(tmp_path / 'one.ckpt')
    a.save_checkpoint(path)
    b.set_state(**a.get_state())
    c.load_checkpoint(path)
    for e in (b, c):
        assert (e.step_num, e.ep_no) == (a.step_num, 41)
        assert np.array_equal(e.obs_image, a.obs_image) and np.array_equal(e.desired_goal, a.desired_goal) and np.array_equal(e.INIT_OBS, a.INIT_OBS)
        assert np.array_equal(e.achieved_goal_vector, a.achieved_goal_vector) and np.array_equal(e.desired_goal_vector, a.desired_goal_vector)
        assert np.array_equal(e.obs_one_hot, a.obs_one_hot) and np.array_equal(e.INIT_OBS_VECTOR, a.INIT_OBS_VECTOR) and e.agent_pos == a.agent_pos
    n_resets = 0
    for t in range(150):
        act = int(pol.randint(6))
        ra, rb, rc = a.step(act), b.step(act), c.step(act)
        for r in (rb, rc):
            assert r[1:3] == ra[1:3] and np.array_equal(frame(r[0]), frame(ra[0])) and np.array_equal(r[3]['achieved_goal'], ra[3]['achieved_goal']), t
        if ra[2]:
            oa, ob, oc = a.reset(), b.reset(), c.reset()
            n_resets += 1
            for o in (ob, oc):
                assert np.array_equal(frame(o), frame(oa)) and a.ep_no == b.ep_no == c.ep_no == 41 + n_resets
                if isinstance(o, dict):
                    assert all(np.array_equal(o[k], oa[k]) for k in oa)
    assert n_resets >= 3
    ka, pa = a.get_rng_state()
    for e in (b, c):
        k, p = e.get_rng_state()
        assert p == pa and np.array_equal(k, ka)
    with pytest.raises(ValueError):
        cls(seed=1, size=(6, 6), max_steps=35).load_checkpoint(path)           # another configuration
    for e in (a, b, c):
        e.close()


def test_obs_one_hot_stays_live_on_the_engine():
    """ray.py:119, 326-327: the array a caller read once follows the env (both step paths, both dtypes), and step_num assigned like the reference's
    plain attribute decides `done` on the card"""
    import gym_craftingworld_amd as cw
    for resident in (True, False):
        for ref_dt in (False, True):
            env = cw.CraftingWorldEnv(size=(5, 5), max_steps=30, seed=8, reference_dtypes=ref_dt, resident=resident)
            env.reset()
            oh = env.obs_one_hot
            assert oh is env.obs_one_hot and env.observation_vector['observation'] is oh and np.array_equal(oh, env.INIT_OBS_VECTOR)
            for a in (0, 1, 2, 3, 4, 5, 0, 1):
                env.step(a)
                st = env.get_state()
                r, c = env.agent_pos
                assert oh[r, c, 8] == 1 and oh[:, :, 8].sum() == 1
                assert np.array_equal((oh[:, :, :8] * np.arange(1, 9)).sum(2), st['grid'][0]) and oh[r, c, 9:].sum() == (1 if st['hold'][0] else 0)
            env.step_num = env.MAX_STEPS - 1
            assert env.step(0)[2] is True and env.step_num == env.MAX_STEPS
            env.close()


@pytest.mark.parametrize('cls_name', ['CraftingWorldEnv', 'CraftingWorldEnvOneHot', 'CraftingWorldEnvFlat', 'CraftingWorldEnvAltObs'])
def test_deepcopy_gives_an_independent_twin(cls_name):
    """copy.deepcopy(env), as planners do with the reference's plain-Python env: the twin continues exactly like the original would (same observations,
    rewards, resets -- the RNG stream, the fixed_init_state pool and the counters travel), stepping one does not move the other, and a copy taken before the
    first reset() resets to the same first episode."""
    import copy
    import gym_craftingworld_amd as cw
    cls = getattr(cw, cls_name)
    kw = dict(size=(6, 6), max_steps=25)
    if cls_name != 'CraftingWorldEnvFlat':
        kw['fixed_init_state'] = 3
    a = cls(seed=17, **kw)
    early = copy.deepcopy(a)                                 # before the first reset

    def frame(o):
        return o if isinstance(o, np.ndarray) else o['observation']
    o1, o2 = a.reset(), early.reset()
    assert np.array_equal(frame(o1), frame(o2)) and np.array_equal(a.desired_goal_vector, early.desired_goal_vector)
    pol = np.random.RandomState(3)
    for _ in range(11):
        if a.step(int(pol.randint(6)))[2]:
            a.reset()
    a.ep_no = 9
    b = copy.deepcopy(a)
    assert type(b) is cls and b is not a and b._eng.value != a._eng.value and (b.step_num, b.ep_no) == (a.step_num, 9)
    assert np.array_equal(b.obs_image, a.obs_image) and np.array_equal(b.desired_goal, a.desired_goal) and np.array_equal(b.obs_one_hot, a.obs_one_hot)
    snap = frame(a._obs_dict() if cls_name != 'CraftingWorldEnvFlat' else a.obs_image).copy()
    for _ in range(5):                                       # the twin walks off: the original does not move
        b.step(int(pol.randint(4)))
    assert np.array_equal(frame(a._obs_dict() if cls_name != 'CraftingWorldEnvFlat' else a.obs_image), snap)
    c = copy.deepcopy(a)
    acts = [int(x) for x in pol.randint(0, 6, size=90)]
    n_resets = 0
    for t, act in enumerate(acts):
        ra, rc = a.step(act), c.step(act)
        assert ra[1:3] == rc[1:3] and np.array_equal(frame(ra[0]), frame(rc[0])), t
        if ra[2]:
            oa, oc = a.reset(), c.reset()
            n_resets += 1
            assert np.array_equal(frame(oa), frame(oc)) and a.ep_no == c.ep_no
    assert n_resets >= 2
    for e in (a, b, c, early):
        e.close()
