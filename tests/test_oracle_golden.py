"""Pin the CPU oracle to the reference: replay every golden fixture (captured by running the
reference itself, tools/gen_golden.py) through oracle/cw_oracle.c and require bit-equality of
everything the reference exposed -- reward, done, achieved/desired bits, agent, hold, grid,
observation image, desired-goal image, init image and the MT19937 stream position."""
import numpy as np
import pytest

from golden_util import crc, fixture_names, load, one_hot
from oracle import OracleEnv


@pytest.mark.parametrize('name', fixture_names())
def test_oracle_matches_reference_fixture(name):
    meta, kw, g = load(name)
    env = OracleEnv(rng_state=(g['key0'], int(g['pos0'])), alt_obs=(meta['env'] == 'CraftingWorldEnvAltObs'), **kw)
    size = kw['size'][0]
    ri = 0
    # CraftingWorldEnvOneHot fixtures (carftingworld_onehot.py): what the class returned are (S,S,12) one-hot STATES, not
    # frames -- observation = current state (:369-371), desired_goal = imagine_obs' final state un-rendered (:310),
    # init_observation = the state at reset (:203).  CraftingWorldEnvFlat fixtures hold the bare frame it returned
    # (craftingworld_flat.py:119,185) = the oracle's observation frame.
    onehot = meta['env'] == 'CraftingWorldEnvOneHot'

    def views(obs, s):
        if not onehot:
            return obs['observation'], obs['desired_goal'], obs['init_observation']
        return (one_hot(s['grid'], s['agent'], s['hold']), one_hot(s['goal_grid'], s['goal_agent']),
                one_hot(s['init_grid'], s['init_agent']))

    def check_reset(t):
        nonlocal ri
        obs = env.reset()
        s = env.state()
        assert s['desired'] == g['r_desired'][ri], (name, 'desired', ri)
        assert np.array_equal(s['grid'], g['r_grid'][ri]), (name, 'reset grid', ri)
        assert s['agent'] == tuple(g['r_agent'][ri])
        key, pos = env.get_rng()
        assert pos == g['r_rng_pos'][ri], (name, 'rng pos', ri)
        assert crc(key) == g['r_rng_crc'][ri]
        o_obs, o_goal, o_init = views(obs, s)
        assert crc(o_obs) == g['r_obs_crc'][ri]
        assert crc(o_goal) == g['r_desired_img_crc'][ri], (name, 'desired_goal', ri)
        assert crc(o_init) == g['r_init_img_crc'][ri]
        assert g['r_at_step'][ri] == t
        assert s['ep_no'] == g['r_ep_no'][ri]
        if onehot:
            assert np.array_equal(s['goal_grid'], g['r_goal_grid'][ri]) and s['goal_agent'] == tuple(g['r_goal_agent'][ri]), (name, 'goal state', ri)
        if ri < len(g['img_desired']):
            assert np.array_equal(o_goal, g['img_desired'][ri])
            assert np.array_equal(o_obs, g['img_obs'][ri])
        ri += 1

    check_reset(0)
    for t, a in enumerate(g['action']):
        obs, r, d, info = env.step(int(a))
        s = env.state()
        assert r == g['reward'][t], (name, 'reward', t)
        assert d == bool(g['done'][t]), (name, 'done', t)
        assert s['achieved'] == g['achieved'][t], (name, 'achieved', t, bin(s['achieved']), bin(g['achieved'][t]))
        assert s['agent'] == tuple(g['agent'][t]), (name, 'agent', t)
        assert s['hold'] == g['hold'][t], (name, 'hold', t)
        assert s['step_num'] == g['step_num'][t]
        assert crc(s['grid']) == g['grid_crc'][t], (name, 'grid', t)
        assert crc(views(obs, s)[0]) == g['obs_crc'][t], (name, 'obs', t)
        if size <= 8:
            assert np.array_equal(s['grid'], g['grid'][t])
        if d:
            check_reset(t + 1)
    assert ri == len(g['r_desired'])
    s = env.state()
    assert np.array_equal(views({'observation': s['obs'], 'desired_goal': None, 'init_observation': None}, s)[0], g['final_obs'])
