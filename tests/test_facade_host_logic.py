"""CPU tier: the HOST logic of the N=1 facade classes (gym_craftingworld_amd/env.py) with the engine replaced by tests/fake_engine.py (the oracle
steps; nothing here claims GPU parity -- tests/test_hip_parity.py replays the same fixtures through the HIP engine).  What is pinned: the
reference-visible OBJECT behaviour captured from the reference class itself in tests/golden/ray5_alias.npz (tools/gen_golden.py: np_random as the
live generator, goal vectors rebound by reset(), negative action ids), and the facade's own contracts around it."""
import numpy as np
import pytest

import fake_engine
from golden_util import (A_CHECK_KEPT, ALIAS_KEPT_OBS_COL, alias_script, crc, load, run_alias_script)


def _alias_env(cls, g, kw, **extra):
    env = cls(**kw, **extra)
    env.set_rng_state(g['key0'], int(g['pos0']))
    return env


@pytest.mark.parametrize('reference_dtypes', [False, True])
@pytest.mark.parametrize('resident', [True, False])
@pytest.mark.parametrize('name', ['ray5_alias', 'flat5_alias', 'onehot5_alias'])
def test_alias_fixture_replays_through_the_facade(monkeypatch, name, resident, reference_dtypes):
    """each fixture was captured from the reference class of that name (Ray: a dict of images; Flat: the bare frame; OneHot: a dict of one-hot states)"""
    fake_engine.install(monkeypatch, resident=resident)
    import gym_craftingworld_amd as cw
    meta, kw, g = load(name)
    ops, args = alias_script()
    assert np.array_equal(ops, g['ops']) and np.array_equal(args, g['args']), 'the fixture was captured with another script: regenerate it'
    env = _alias_env(getattr(cw, meta['env']), g, kw, reference_dtypes=reference_dtypes)
    rows = run_alias_script(env, ops, args, meta['policy_seed'])
    want = g['rows'].copy()
    if not reference_dtypes:          # default uint8 frames are the engine's live buffers: a kept observation shows the NEW episode after reset() (documented)
        keep = ops == A_CHECK_KEPT
        assert (rows[keep, ALIAS_KEPT_OBS_COL] != want[keep, ALIAS_KEPT_OBS_COL]).any()
        rows[keep, ALIAS_KEPT_OBS_COL] = want[keep, ALIAS_KEPT_OBS_COL]
    bad = np.nonzero((rows != want).any(axis=1))[0]
    assert bad.size == 0, 'op %d (%d, arg %d): got %s, the reference %s' % (bad[0], ops[bad[0]], args[bad[0]], rows[bad[0]].tolist(), want[bad[0]].tolist())
    env.close()


def test_np_random_mirror_moves_no_state_unless_touched(monkeypatch):
    """the mirror is lazy: a reset / step loop that never looks at np_random uploads and downloads nothing; one look costs one download; one draw
    costs one upload at the next reset()"""
    fake_engine.install(monkeypatch)
    import gym_craftingworld_amd as cw
    env = cw.CraftingWorldEnv(size=(5, 5), max_steps=10, seed=3)
    v = env._vec
    up0, down0 = v.n_rng_uploads, v.n_rng_downloads
    for _ in range(3):
        env.reset()
        for a in range(6):
            env.step(a)
    assert (v.n_rng_uploads, v.n_rng_downloads) == (up0, down0)
    pos = env.np_random.get_state()[2]                  # a look: one download, nothing to upload later
    assert (v.n_rng_uploads, v.n_rng_downloads) == (up0, down0 + 1)
    env.reset()
    assert (v.n_rng_uploads, v.n_rng_downloads) == (up0, down0 + 1)
    assert env.np_random.get_state()[2] != pos or True  # (the position moved or wrapped; what matters is the traffic)
    assert v.n_rng_downloads == down0 + 2
    env.np_random.randint(10)                           # a draw: uploaded before the engine's next draw, once
    env.step(0)
    assert v.n_rng_uploads == up0
    env.reset()
    assert v.n_rng_uploads == up0 + 1
    env.reset()
    assert v.n_rng_uploads == up0 + 1
    env.close()


def test_np_random_is_a_randomstate_and_pickles_detached(monkeypatch):
    import copy
    import pickle
    fake_engine.install(monkeypatch)
    import gym_craftingworld_amd as cw
    env = cw.CraftingWorldEnv(size=(5, 5), max_steps=10, seed=11)
    env.reset()
    rs = env.np_random
    assert isinstance(rs, np.random.RandomState) and rs is env.np_random
    clone = pickle.loads(pickle.dumps(rs))
    assert type(clone) is np.random.RandomState
    dc = copy.deepcopy(rs)
    want = [int(x) for x in clone.randint(0, 1 << 30, size=5)]
    assert [int(x) for x in dc.randint(0, 1 << 30, size=5)] == want
    k, p = env.get_rng_state()                      # drawing from the copies did not move the env's stream
    again = np.random.RandomState()
    again.set_state(('MT19937', k, p, 0, 0.0))
    assert [int(x) for x in again.randint(0, 1 << 30, size=5)] == want
    with pytest.raises(ValueError):
        env.np_random = np.random.default_rng(1)    # gym >= 0.22's Generator: not what the reference's reset() can use (.randint)
    env.close()


def test_negative_action_ids_and_the_errors_around_them(monkeypatch):
    fake_engine.install(monkeypatch)
    import gym_craftingworld_amd as cw
    a = cw.CraftingWorldEnv(size=(5, 5), max_steps=40, seed=5)
    b = cw.CraftingWorldEnv(size=(5, 5), max_steps=40, seed=5)
    with pytest.raises(TypeError):
        a.step(-6)                                      # before reset: `None + Coord` (a move), ray.py:393
    with pytest.raises(AttributeError):
        a.step(-1)                                      # `None.tuple()` (drop), ray.py:330
    assert a.step_num == 2
    a.step_num = 0
    a.reset(), b.reset()
    pol = np.random.RandomState(0)
    for _ in range(120):
        n = int(pol.randint(-6, 0))
        oa, ra, da, _ = a.step(n)
        ob, rb, db, _ = b.step(n + 6)
        assert (ra, da) == (rb, db) and np.array_equal(oa['observation'], ob['observation'])
        if da:
            a.reset(), b.reset()
    for bad in (6, -7, 100):
        with pytest.raises(IndexError):
            a.step(bad)
    a.close(), b.close()


def test_flat_rejects_fixed_init_state_like_the_reference(monkeypatch):
    fake_engine.install(monkeypatch)
    import gym_craftingworld_amd as cw
    with pytest.raises(TypeError, match='fixed_init_state'):
        cw.CraftingWorldEnvFlat(fixed_init_state=2)                               # craftingworld_flat.py:52-55 has no such kwarg
    with pytest.raises(TypeError):
        cw.CraftingWorldEnvFlat(bogus=1)
    env = cw.CraftingWorldEnvFlat(seed=1, reference_dtypes=True)
    assert env.STATE_W == 8 and env.MAX_STEPS == 100 and env.reset().shape == (32, 32, 3)
    env.close()


def test_state_round_trip_and_written_through_counters(monkeypatch):
    """get_state() / set_state() on the N=1 class: a second env given the first one's state continues exactly like it; step_num is
    written through to the engine (done is decided there); ep_no travels with the state"""
    fake_engine.install(monkeypatch)
    import gym_craftingworld_amd as cw
    kw = dict(size=(6, 6), max_steps=25)
    a, b = cw.CraftingWorldEnv(seed=21, **kw), cw.CraftingWorldEnv(seed=99, **kw)
    with pytest.raises(RuntimeError):
        a.get_state()
    a.reset(), b.reset()
    pol = np.random.RandomState(4)
    for _ in range(9):
        a.step(int(pol.randint(6)))
    a.ep_no = 7
    st = a.get_state()
    assert st['grid'].shape == (1, 6, 6) and st['rng_key'].shape == (1, 624) and int(st['ep_no'][0]) == 7 and int(st['step_num'][0]) == 9
    sub = {k: st[k] for k in ('grid', 'init_grid', 'agent_rc', 'hold', 'achieved', 'desired', 'step_num', 'ep_no', 'rng_key', 'rng_pos')}
    b.set_state(**sub)
    assert b.step_num == 9 and b.ep_no == 7 and np.array_equal(b.obs_image, a.obs_image)
    assert np.array_equal(b.achieved_goal_vector, a.achieved_goal_vector) and np.array_equal(b.obs_one_hot, a.obs_one_hot)
    for t in range(40):
        act = int(pol.randint(6))
        oa, ra, da, ia = a.step(act)
        ob, rb, db, ib = b.step(act)
        assert (ra, da) == (rb, db) and np.array_equal(oa['observation'], ob['observation']) and np.array_equal(ia['achieved_goal'], ib['achieved_goal']), t
        if da:
            oa, ob = a.reset(), b.reset()
            # (the goal image of the RUNNING episode is not restorable by the fake engine; from the next reset on everything is)
            assert np.array_equal(oa['desired_goal'], ob['desired_goal']) and a.ep_no == b.ep_no == 8
    # fields without the leading dimension are taken too
    b.set_state(step_num=3, agent_rc=np.array([2, 2]))
    assert b.step_num == 3 and b.agent_pos == (2, 2)
    with pytest.raises(ValueError):
        b.set_state(rng_key=st['rng_key'])
    with pytest.raises(ValueError):
        b.set_state(colour=1)
    # step_num assigned like a plain attribute of the reference: the episode ends when the ENGINE's count says so
    a.reset()
    a.step_num = a.MAX_STEPS - 1
    assert a.step(0)[2] is True and a.step_num == a.MAX_STEPS
    a.close(), b.close()


def test_obs_one_hot_is_one_live_array_per_episode(monkeypatch):
    """ray.py:119, 179, 326-327: obs_one_hot is ONE array that later steps mutate -- from its first read on, on both step paths"""
    for resident in (True, False):
        fake_engine.install(monkeypatch, resident=resident)
        import gym_craftingworld_amd as cw
        for ref_dt in (False, True):
            env = cw.CraftingWorldEnv(size=(5, 5), max_steps=30, seed=8, reference_dtypes=ref_dt)
            assert env.obs_one_hot is None
            env.reset()
            oh = env.obs_one_hot
            assert oh is env.obs_one_hot and oh.shape == (5, 5, 12) and oh.dtype == (np.int64 if ref_dt else np.uint8)
            assert env.observation_vector['observation'] is oh and env.INIT_OBS_VECTOR is env.INIT_OBS_VECTOR
            assert np.array_equal(oh, env.INIT_OBS_VECTOR)
            moved = False
            for a in (0, 1, 2, 3, 0, 1):
                before = oh.copy()
                env.step(a)
                r, c = env.agent_pos
                assert oh[r, c, 8] == 1 and oh[:, :, 8].sum() == 1          # the held reference follows the agent
                moved = moved or not np.array_equal(before, oh)
            assert moved
            init = env.INIT_OBS_VECTOR
            env.reset()
            assert env.INIT_OBS_VECTOR is not init
            assert (env.obs_one_hot is oh) == (not ref_dt)                   # int64 copies are per-episode arrays like the reference's; the uint8 view is the engine's buffer
            env.close()


def test_an_assigned_generator_is_uploaded_only_when_the_caller_moved_it(monkeypatch):
    """a plain RandomState assigned to env.np_random cannot be hooked: the env looks at it before every draw of the engine and writes it back after --
    one download per reset(), and an upload only when the caller has drawn from it (or set it) since"""
    fake_engine.install(monkeypatch)
    import gym_craftingworld_amd as cw
    env = cw.CraftingWorldEnv(size=(5, 5), max_steps=10, seed=3)
    v = env._vec
    rs = np.random.RandomState(77)
    env.np_random = rs
    assert env.np_random is rs
    up0, down0 = v.n_rng_uploads, v.n_rng_downloads
    for i in range(4):
        env.reset()
        assert (v.n_rng_uploads, v.n_rng_downloads) == (up0, down0 + i + 1)
    ref = np.random.RandomState()
    ref.set_state(rs.get_state())
    a = rs.randint(1000)                                 # the caller draws: the env's next reset() starts behind that draw
    assert a == ref.randint(1000)
    env.reset()
    assert v.n_rng_uploads == up0 + 1
    k, p = env.get_rng_state()
    assert p == rs.get_state()[2] and np.array_equal(k, rs.get_state()[1])
    rs.seed(5)
    env.reset()
    assert v.n_rng_uploads == up0 + 2
    env.close()
