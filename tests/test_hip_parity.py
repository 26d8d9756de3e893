"""GPU parity tests: the HIP engine (through the C ABI, via gym_craftingworld_amd) against
  (1) the golden fixtures captured from the reference itself (tests/golden, tools/gen_golden.py),
  (2) the CPU oracle on seeded random batches (auto-reset, mixed task menus, 5x5..32x32),
  (3) injected single-step states covering the whole transition table,
  (4) size-independent properties at BASELINE's full batch size (65 536 envs).
Bit-exact everywhere: the path is integer-only.  NOTHING here asserts a rate, a roofline fraction, a guard move or a duration: those live in
tests/test_zz_perf_floors.py, collected last, so that `pytest -x` on a slow box cannot hide a parity test behind a timing failure.  The
strongest test -- every env of the full-size batches against the oracle -- comes first."""
import os
import numpy as np
import pytest
import torch


from golden_util import crc, fixture_names, load


pytestmark = pytest.mark.gpu


TASKS = ['MakeBread', 'EatBread', 'BuildHouse', 'ChopTree', 'ChopRock', 'GoToHouse', 'MoveAxe', 'MoveHammer',
         'MoveSticks']


def _np_states(n, base):
    sts = [np.random.RandomState(base + i).get_state() for i in range(n)]
    keys = np.stack([s[1] for s in sts]).astype(np.uint32)
    pos = np.array([s[2] for s in sts], dtype=np.int32)
    return keys, pos


def _hdr_fields(hdr):
    h = hdr.cpu().numpy().astype(np.int64)
    return dict(agent=h[:, 0:2], hold=h[:, 2], achieved=h[:, 4] | (h[:, 5] << 8), desired=h[:, 6] | (h[:, 7] << 8),
                step_num=h[:, 8] | (h[:, 9] << 8))


# ------------------------------------------------------------------ (0) every env of the full-size batches: first, so that no later failure hides it


_MENUS8 = [dict(), dict(selected_tasks=TASKS[::-1]), dict(selected_tasks=TASKS[:4], number_of_tasks=2),
           dict(selected_tasks=['GoToHouse', 'MoveAxe', 'EatBread'], stacking=False), dict(selected_tasks=TASKS[3:], reward_style='subset'),
           dict(selected_tasks=['ChopTree', 'BuildHouse'], number_of_tasks=1), dict(selected_tasks=TASKS[1::2]),
           dict(selected_tasks=TASKS[::2], number_of_tasks=3, reward_style='subset')]


@pytest.mark.gpu
@pytest.mark.parametrize('N,size,T,menus,raster', [(65536, 21, 80, False, 'ray'), (65536, 32, 48, False, 'ray'), (131072, 21, 60, True, 'ray'),
                                                    (65536, 21, 60, False, 'alt'), (65536, 8, 60, False, 'ray'), (65536, 5, 60, False, 'ray')],
                         ids=['configs2_65536x21', 'configs4_65536x32', 'configs3_131072_mixed_menus', 'altobs_65536x21', 'flat_default_65536x8', 'gather_65536x5'])
def test_every_env_of_the_full_size_batches_against_the_oracle(N, size, T, menus, raster):
    """BASELINE configs[2], [4] and [3]'s per-GPU share at FULL size with EVERY env checked against the CPU oracle, not a sample (the round-4 verdict's
    caveat): full frames, auto-reset, pre-generated random actions, episodes of 37 steps with the phases spread out (envs finish on every step, at least
    once each on the way).  The engine records reward and done of every step on the device; the oracle then replays the same actions in slices of
    8 192 envs on all host threads (cwo_batch_rollout) and every reward, every done, and at the end every env's three frames, state and RNG stream must
    be the engine's.  Also at that size: the AltObs raster, the Flat id's 8x8 default (four workgroups per CU) and 5x5 (the gather painter)."""
    from gym_craftingworld_amd import CraftingWorldVecEnv
    from oracle import OracleBatch
    SL = 8192
    kw = dict(size=(size, size), max_steps=37)
    env_menu = (np.arange(N) % 8).astype(np.uint8) if menus else None
    env = CraftingWorldVecEnv(N, obs_mode='pixels', seed=77, raster=raster, **kw, **(dict(task_menus=_MENUS8, env_menu=env_menu) if menus else {}))
    if raster == 'alt':
        kw['alt_obs'] = True                                 # (the oracle's name for CraftingWorldEnvAltObs's rasteriser)
    keys, pos = env.get_rng_states()
    phase = (np.arange(N) % 31).astype(np.int32)
    obs = env.reset()
    env.set_state(step_num=phase)
    gen = torch.Generator(device='cuda').manual_seed(21)
    acts = torch.randint(0, 6, (T, N), device='cuda', dtype=torch.uint8, generator=gen)
    rec_r = torch.empty((T, N), dtype=torch.int32, device='cuda')
    rec_d = torch.empty((T, N), dtype=torch.bool, device='cuda')
    for t in range(T):
        obs, r, d, _ = env.step(acts[t])
        rec_r[t] = r
        rec_d[t] = d
    torch.cuda.synchronize()
    a_host, r_host, d_host = acts.cpu().numpy().astype(np.int8), rec_r.cpu().numpy(), rec_d.cpu().numpy()
    st = env.get_state()
    k2, p2 = env.get_rng_states()
    threads = max(1, len(os.sched_getaffinity(0)))
    finished = 0
    ra, rb = np.random.RandomState(), np.random.RandomState()
    for lo in range(0, N, SL):
        hi = lo + SL
        ora = OracleBatch(SL, rng_states=[(keys[i], int(pos[i])) for i in range(lo, hi)],
                          per_env_kwargs=[_MENUS8[int(m)] for m in env_menu[lo:hi]] if menus else None, **kw)
        ora.reset()
        for j, e in enumerate(ora.envs):                     # the same phase spread (step_num only)
            v = e.view()
            e._lib.cwo_set_state(e._h, v.grid, v.init_grid, v.agent_r, v.agent_c, v.hold, v.achieved, v.desired, int(phase[lo + j]))
        total, o_rew, o_done = ora.rollout(a_host[:, lo:hi], nthreads=threads, record=True)
        assert total == SL * T
        assert np.array_equal(r_host[:, lo:hi], o_rew), ('reward', lo)
        assert np.array_equal(d_host[:, lo:hi], o_done.astype(bool)), ('done', lo)
        finished += int(o_done.sum())
        f_obs, f_goal, f_init = (obs[k][lo:hi].cpu().numpy() for k in ('observation', 'desired_goal', 'init_observation'))
        ish = ora.envs[0].img_shape
        for j, e in enumerate(ora.envs):
            v, i = e.view(), lo + j
            assert np.array_equal(f_obs[j], np.ctypeslib.as_array(v.obs, shape=ish)), ('observation', i)
            assert np.array_equal(f_goal[j], np.ctypeslib.as_array(v.desired_img, shape=ish)), ('desired_goal', i)
            assert np.array_equal(f_init[j], np.ctypeslib.as_array(v.init_img, shape=ish)), ('init_observation', i)
            assert (st['agent_rc'][i][0], st['agent_rc'][i][1], st['hold'][i], st['achieved'][i], st['desired'][i], st['step_num'][i]) == \
                (v.agent_r, v.agent_c, v.hold, v.achieved, v.desired, v.step_num), ('state', i)
            assert np.array_equal(st['grid'][i].reshape(-1), np.ctypeslib.as_array(v.grid, shape=(size * size,))), ('grid', i)
            ok, op = e.get_rng()                             # (the same point of the same stream: numpy holds 624 where the engine holds 0 -- compare what comes next)
            assert op % 624 == int(p2[i]) % 624, ('rng position', i)
            if j % 8 == 0:
                ra.set_state(('MT19937', ok, op, 0, 0.0))
                rb.set_state(('MT19937', k2[i], int(p2[i]), 0, 0.0))
                assert np.array_equal(ra.randint(0, 2**32, 4, dtype=np.uint32), rb.randint(0, 2**32, 4, dtype=np.uint32)), ('rng stream', i)
        del ora
    assert finished == int(env.counters[1].item()) and finished >= N
    env.close()


@pytest.mark.gpu
@pytest.mark.parametrize('obs_mode', ['pixels_dirty', 'state'])
def test_every_env_of_a_batch_whose_episodes_end_early_against_the_oracle(obs_mode):
    """What a policy that SUCCEEDS does to the reset path (round 6): 65 536 envs on 8x8 grids, reward_style='subset', the one task EatBread, a
    random walker -- episodes of ~140 steps under max_steps = 300, ~480 envs finishing on every step, many of them two, three, four times between two
    look-ahead refills (the ring of records runs down, the refill period adapts, a few are reset the slow way).  Every reward and done of 450 steps, and at
    the end every env's state, frames, episode counters and RNG state (key and position, exactly), against the oracle."""
    from gym_craftingworld_amd import CraftingWorldVecEnv
    from oracle import OracleBatch
    N, T, SL, size = 65536, 450, 8192, 8
    kw = dict(size=(size, size), max_steps=300, reward_style='subset', selected_tasks=['EatBread'], number_of_tasks=1)
    env = CraftingWorldVecEnv(N, obs_mode=obs_mode, seed=123, **kw)
    keys, pos = env.get_rng_states()
    obs = env.reset()
    acts = torch.randint(0, 4, (T, N), device='cuda', dtype=torch.uint8, generator=torch.Generator(device='cuda').manual_seed(9))
    rec_r = torch.empty((T, N), dtype=torch.int32, device='cuda')
    rec_d = torch.empty((T, N), dtype=torch.bool, device='cuda')
    for t in range(T):
        obs, r, d, _ = env.step(acts[t])
        rec_r[t] = r
        rec_d[t] = d
    torch.cuda.synchronize()
    a_host, r_host, d_host = acts.cpu().numpy().astype(np.int8), rec_r.cpu().numpy(), rec_d.cpu().numpy()
    st = env.get_state()
    k2, p2 = env.get_rng_states()
    threads = max(1, len(os.sched_getaffinity(0)))
    finished, most = 0, 0
    for lo in range(0, N, SL):
        hi = lo + SL
        ora = OracleBatch(SL, rng_states=[(keys[i], int(pos[i])) for i in range(lo, hi)], **kw)
        ora.reset()
        total, o_rew, o_done = ora.rollout(a_host[:, lo:hi], nthreads=threads, record=True)
        assert total == SL * T
        assert np.array_equal(r_host[:, lo:hi], o_rew), ('reward', lo)
        assert np.array_equal(d_host[:, lo:hi], o_done.astype(bool)), ('done', lo)
        finished += int(o_done.sum())
        per_env = o_done.sum(axis=0)
        most = max(most, int(np.add.reduceat(o_done[:448], np.arange(0, 448, 64), axis=0).max()))      # most finishes of one env inside 64 consecutive steps
        if obs_mode != 'state':
            f_obs, f_goal, f_init = (obs[k][lo:hi].cpu().numpy() for k in ('observation', 'desired_goal', 'init_observation'))
        ish = ora.envs[0].img_shape
        for j, e in enumerate(ora.envs):
            v, i = e.view(), lo + j
            if obs_mode != 'state':
                assert np.array_equal(f_obs[j], np.ctypeslib.as_array(v.obs, shape=ish)), ('observation', i)
                assert np.array_equal(f_goal[j], np.ctypeslib.as_array(v.desired_img, shape=ish)), ('desired_goal', i)
                assert np.array_equal(f_init[j], np.ctypeslib.as_array(v.init_img, shape=ish)), ('init_observation', i)
            assert (st['agent_rc'][i][0], st['agent_rc'][i][1], st['hold'][i], st['achieved'][i], st['desired'][i], st['step_num'][i], st['ep_no'][i]) == \
                (v.agent_r, v.agent_c, v.hold, v.achieved, v.desired, v.step_num, v.ep_no), ('state', i)
            assert np.array_equal(st['grid'][i].reshape(-1), np.ctypeslib.as_array(v.grid, shape=(size * size,))), ('grid', i)
            ok, op = e.get_rng()
            assert op == int(p2[i]) and np.array_equal(ok, k2[i]), ('rng state', i, int(per_env[j]))
        del ora
    c = env._counters_raw.cpu()
    assert finished == int(c[1]) and 0.8 * finished < int(c[2]) <= finished and finished > 2 * N      # (mostly successes -- the rest walked 300 steps --; each env finished ~3 times)
    assert most >= 5                                                         # (some env ran its ring of four records empty between two refills ...)
    assert 0 < int(c[5]) < finished // 100                                   # (... and was reset the slow way: rare, and it changes nothing)
    env.close()


# ------------------------------------------------------------------ (1) golden fixtures
@pytest.mark.parametrize('obs_mode', ['pixels', 'pixels_dirty'])
@pytest.mark.parametrize('name', fixture_names())
def test_golden_fixture_replay(name, obs_mode):
    from gym_craftingworld_amd import CraftingWorldVecEnv
    meta, kw, g = load(name)
    env = CraftingWorldVecEnv(1, obs_mode=obs_mode, seed=0, raster='alt' if meta['env'] == 'CraftingWorldEnvAltObs' else 'ray', **kw)
    env.set_rng_states(g['key0'][None], np.array([g['pos0']]))
    if kw.get('fixed_init_state'):
        # the pool is drawn from the env RNG at construction (ray.py:116-118): redo it on the injected stream
        import ctypes as C
        from gym_craftingworld_amd import _lib as L
        L.check(env._lib.cw_generate_fixed_states(env._h, env._stream()), 'pool')
    size = kw['size'][0]
    ri = 0
    # fixtures captured from CraftingWorldEnvOneHot hold what THAT class returns: (S,S,12) one-hot states
    # (carftingworld_onehot.py:203,310,369-371) -- the engine's one-hot views of the current / goal / reset states
    onehot = meta['env'] == 'CraftingWorldEnvOneHot'

    def view(obs, key):
        if onehot:
            which = {'observation': 'current', 'desired_goal': 'goal', 'init_observation': 'init'}[key]
            return env.one_hot(which=which)[0].cpu().numpy()
        return obs[key][0].cpu().numpy()

    def check_reset(obs, t):
        nonlocal ri
        st = env.get_state()
        assert st['desired'][0] == g['r_desired'][ri], (name, 'desired', ri)
        assert np.array_equal(st['grid'][0], g['r_grid'][ri]), (name, 'reset grid', ri)
        assert tuple(st['agent_rc'][0]) == tuple(g['r_agent'][ri])
        assert crc(view(obs, 'observation')) == g['r_obs_crc'][ri], (name, 'reset obs', ri)
        assert crc(view(obs, 'desired_goal')) == g['r_desired_img_crc'][ri], (name, 'desired img', ri)
        assert crc(view(obs, 'init_observation')) == g['r_init_img_crc'][ri]
        assert g['r_at_step'][ri] == t
        assert st['ep_no'][0] == g['r_ep_no'][ri]
        # the MT19937 state after EVERY reset is the reference generator's own: RandomState.get_state()'s key and position (cw_get_mt reports numpy's form)
        k_, p_ = env.get_rng_states()
        assert p_[0] == g['r_rng_pos'][ri] and crc(k_[0]) == g['r_rng_crc'][ri], (name, 'rng state', ri)
        if onehot:
            assert np.array_equal(st['goal_grid'][0], g['r_goal_grid'][ri]) and tuple(st['goal_agent_rc'][0]) == tuple(g['r_goal_agent'][ri])
        if ri < len(g['img_desired']):
            assert np.array_equal(view(obs, 'desired_goal'), g['img_desired'][ri])
        ri += 1

    obs = env.reset()
    check_reset(obs, 0)
    T = len(g['action'])
    acts = torch.as_tensor(g['action'].astype(np.int32), device=env.device)
    check_every = 1 if T <= 3000 else 2
    for t in range(T):
        obs, rew, done, info = env.step(acts[t:t + 1])
        r, d = int(rew[0].item()), bool(done[0].item())
        assert r == g['reward'][t], (name, 'reward', t)
        assert d == bool(g['done'][t]), (name, 'done', t)
        ach = int(info['achieved_goal'][0].item()) & 0xFFFF
        assert ach == g['achieved'][t], (name, 'achieved', t, bin(ach), bin(g['achieved'][t]))
        if d:
            check_reset(obs, t + 1)
        elif t % check_every == 0:
            f = _hdr_fields(env.hdr)
            assert tuple(f['agent'][0]) == tuple(g['agent'][t]), (name, 'agent', t)
            assert f['hold'][0] == g['hold'][t], (name, 'hold', t)
            assert f['step_num'][0] == g['step_num'][t]
            assert crc(view(obs, 'observation')) == g['obs_crc'][t], (name, 'obs', t)
            assert crc(env.grid()[0].cpu().numpy()) == g['grid_crc'][t], (name, 'grid', t)
    assert ri == len(g['r_desired'])
    assert np.array_equal(view(obs, 'observation'), g['final_obs'])
    # the MT19937 stream position after the last reset is the reference's
    keys, pos = env.get_rng_states()
    assert pos[0] == g['r_rng_pos'][-1] and crc(keys[0]) == g['r_rng_crc'][-1]
    env.close()


# ------------------------------------------------------------------ (2) oracle, random batches
CASES = [
    dict(N=256, T=130, kw=dict(size=(5, 5), max_steps=20)),
    dict(N=192, T=650, kw=dict(size=(21, 21), max_steps=300)),
    dict(N=64, T=120, kw=dict(size=(32, 32), max_steps=50)),
    dict(N=128, T=200, kw=dict(size=(6, 6), max_steps=30, reward_style='subset', stacking=False)),
    dict(N=96, T=150, kw=dict(size=(7, 7), max_steps=25, fixed_init_state=4)),
    dict(N=100, T=100, kw=dict(size=(4, 4), max_steps=15, selected_tasks=['GoToHouse', 'EatBread', 'MoveAxe'], number_of_tasks=2)),
    dict(N=65, T=45, kw=dict(size=(64, 64), max_steps=20)),        # one wave-iteration = exactly one grid row
    dict(N=3, T=25, kw=dict(size=(255, 255), max_steps=12)),       # the largest grid the u8/u16 layout admits
    dict(N=33, T=60, kw=dict(size=(13, 13), max_steps=25, task_list=TASKS + ['Extra1', 'Extra2'],
                             selected_tasks=['Extra2', 'ChopTree', 'MoveHammer'])),   # len(task_list) = 11
]


@pytest.mark.parametrize('obs_mode', ['pixels', 'pixels_dirty', 'state'])
@pytest.mark.parametrize('case', CASES, ids=lambda c: 'N%d_S%d' % (c['N'], c['kw']['size'][0]))
def test_random_batch_vs_oracle(case, obs_mode):
    from gym_craftingworld_amd import CraftingWorldVecEnv
    from oracle import OracleBatch
    N, T, kw = case['N'], case['T'], case['kw']
    keys, pos = _np_states(N, 31000)
    # fixed_init_state pools are drawn from the env stream (ray.py:116-118): inject the stream, then redraw the pool
    env = CraftingWorldVecEnv(N, obs_mode=obs_mode, **kw)
    env.set_rng_states(keys, pos)
    if kw.get('fixed_init_state'):
        from gym_craftingworld_amd import _lib as L
        L.check(env._lib.cw_generate_fixed_states(env._h, env._stream()), 'pool')
    ora = OracleBatch(N, rng_states=list(zip(keys, pos)), **kw)
    env.reset()
    ora.reset()
    acts = np.random.RandomState(5).randint(0, 6, size=(T, N)).astype(np.int64)
    dacts = torch.as_tensor(acts, device=env.device)
    n_done = 0
    ret, length = np.zeros(N, np.int64), np.zeros(N, np.int64)         # the oracle's running episode return / length (ray.py:361-367 summed by the loop)
    for t in range(T):
        obs, rew, done, info = env.step(dacts[t])
        # terminal masks must be read before the oracle resets
        o_rew = np.empty(N, np.int32)
        o_done = np.zeros(N, bool)
        o_ach = np.empty(N, np.int64)
        for i, e in enumerate(ora.envs):
            _, o_rew[i], o_done[i], _ = e.step(int(acts[t, i]))
            o_ach[i] = e.view().achieved
            if o_done[i]:
                e.reset()
        assert np.array_equal(rew.cpu().numpy(), o_rew), ('reward', t)
        assert np.array_equal(done.cpu().numpy(), o_done), ('done', t)
        assert np.array_equal(info['achieved_goal'].cpu().numpy().astype(np.int64) & 0xFFFF, o_ach), ('achieved', t)
        ret += o_rew
        length += 1
        if o_done.any():                                 # info['episode'] = {'r', 'l'}: device tensors, rows valid where done
            assert np.array_equal(info['episode']['r'].cpu().numpy()[o_done], ret[o_done]), ('episode return', t)
            assert np.array_equal(info['episode']['l'].cpu().numpy()[o_done], length[o_done]), ('episode length', t)
            assert info['episode']['l'] is info['episode_length'] and info['episode']['r'] is env.episode_return
            ret[o_done] = 0
            length[o_done] = 0
        n_done += int(o_done.sum())
        if t % 37 == 0 or t == T - 1:
            _compare_full(env, ora, obs_mode, t)
    assert n_done > 0
    assert int(env.counters[1].item()) == n_done
    assert int(env.counters[0].item()) == N * T
    # RNG streams continue identically
    k2, p2 = env.get_rng_states()
    for i in (0, N // 2, N - 1):
        rs = np.random.RandomState()
        rs.set_state(('MT19937', k2[i], int(p2[i]), 0, 0.0))
        ok, op = ora.envs[i].get_rng()
        ro = np.random.RandomState()
        ro.set_state(('MT19937', ok, op, 0, 0.0))
        assert np.array_equal(rs.randint(0, 2**32, 700, dtype=np.uint32), ro.randint(0, 2**32, 700, dtype=np.uint32))
    env.close()


def _compare_full(env, ora, obs_mode, t):
    st = env.get_state()
    grid_dev = env.grid().cpu().numpy()
    oh = env.one_hot().cpu().numpy()
    frames = env.render().cpu().numpy()
    if obs_mode != 'state':
        o = env._observation()
        obs, des, ini = (o[k].cpu().numpy() for k in ('observation', 'desired_goal', 'init_observation'))
    for i, s in enumerate(ora.states()):
        tag = ('env', i, 'step', t)
        assert np.array_equal(st['grid'][i], s['grid']), tag
        assert np.array_equal(grid_dev[i], s['grid']), tag
        assert np.array_equal(st['init_grid'][i], s['init_grid']), tag
        assert np.array_equal(st['goal_grid'][i], s['goal_grid']), tag
        assert tuple(st['agent_rc'][i]) == s['agent'] and tuple(st['goal_agent_rc'][i]) == s['goal_agent'], tag
        assert st['hold'][i] == s['hold'] and st['achieved'][i] == s['achieved'] and st['desired'][i] == s['desired'], tag
        assert st['step_num'][i] == s['step_num'] and st['ep_no'][i] == s['ep_no'], tag
        assert np.array_equal(frames[i], s['obs']), tag
        # one-hot view: channels 0-7 objects, 8 agent, 9-11 hold (ray.py:94-98)
        g = s['grid']
        assert np.array_equal(oh[i, :, :, :8].argmax(-1) * oh[i, :, :, :8].any(-1) + oh[i, :, :, :8].any(-1), np.where(g > 0, g, 0)), tag
        assert oh[i, s['agent'][0], s['agent'][1], 8] == 1 and oh[i, :, :, 8].sum() == 1, tag
        assert oh[i, :, :, 9:].sum() == (1 if s['hold'] else 0), tag
        if obs_mode != 'state':
            assert np.array_equal(obs[i], s['obs']), tag
            assert np.array_equal(des[i], s['desired_img']), tag
            assert np.array_equal(ini[i], s['init_img']), tag


def test_mixed_task_menus_vs_oracle():
    """BASELINE config 4 shape: env i uses menu[i % M] (ordered selected_tasks lists)."""
    from gym_craftingworld_amd import CraftingWorldVecEnv
    from oracle import OracleBatch
    menus = [dict(selected_tasks=TASKS), dict(selected_tasks=['MoveSticks', 'MakeBread'], stacking=False),
             dict(selected_tasks=['ChopRock', 'ChopTree', 'BuildHouse', 'GoToHouse'], number_of_tasks=3, reward_style='subset'),
             dict(selected_tasks=list(reversed(TASKS)), number_of_tasks=4)]
    N, T = 128, 160
    env_menu = np.arange(N) % len(menus)
    kw = dict(size=(6, 6), max_steps=25)
    keys, pos = _np_states(N, 777)
    env = CraftingWorldVecEnv(N, obs_mode='pixels_dirty', task_menus=menus, env_menu=env_menu, **kw)
    env.set_rng_states(keys, pos)
    ora = OracleBatch(N, rng_states=list(zip(keys, pos)), per_env_kwargs=[menus[m] for m in env_menu], **kw)
    env.reset()
    ora.reset()
    acts = np.random.RandomState(11).randint(0, 6, size=(T, N)).astype(np.int32)
    dacts = torch.as_tensor(acts, device=env.device)
    for t in range(T):
        obs, rew, done, _ = env.step(dacts[t])
        o_rew, o_done = ora.step(acts[t])
        assert np.array_equal(rew.cpu().numpy(), o_rew) and np.array_equal(done.cpu().numpy(), o_done), t
    _compare_full(env, ora, 'pixels_dirty', T)
    env.close()


# ------------------------------------------------------------------ (3) injected transition table
def test_injected_states_single_step():
    """Random (grid, agent, hold, achieved, desired, init grid) x all 6 actions, one step each, both
    reward styles -- covers blocked moves, failed-move bit flips, Move* 1->0 reversals, every
    object under the agent, edge clamping on all four walls."""
    from gym_craftingworld_amd import CraftingWorldVecEnv
    from oracle import OracleEnv
    rng = np.random.RandomState(2024)
    S, N = 5, 6 * 400
    for style in (None, 'subset'):
        env = CraftingWorldVecEnv(N, size=(S, S), max_steps=10, obs_mode='pixels_dirty', reward_style=style, auto_reset=False)
        env.reset()
        grids = np.zeros((N, S, S), np.uint8)
        inits = np.zeros((N, S, S), np.uint8)
        agent = np.zeros((N, 2), np.uint8)
        hold = np.zeros(N, np.uint8)
        ach = np.zeros(N, np.uint16)
        des = np.zeros(N, np.uint16)
        stepn = np.zeros(N, np.int32)
        for j in range(N // 6):
            cells = rng.permutation(S * S)
            nobj = rng.randint(0, 8)
            g = np.zeros(S * S, np.uint8)
            h = rng.randint(0, 4) if rng.rand() < 0.6 else 0
            g[cells[:nobj]] = rng.randint(1, 9, size=nobj)
            if h and nobj == 8:
                h = 0
            ig = np.zeros(S * S, np.uint8)
            ig[rng.permutation(S * S)[:8]] = np.arange(1, 9)          # one of each, like sample_state
            if rng.rand() < 0.5 and nobj:                              # make "at its origin" cases likely
                ig[:] = 0
                ig[rng.permutation(S * S)[:8]] = np.arange(1, 9)
            a_cell = cells[rng.randint(0, S * S)] if rng.rand() < 0.5 else rng.randint(0, S * S)
            for a in range(6):
                i = j * 6 + a
                grids[i] = g.reshape(S, S)
                inits[i] = ig.reshape(S, S)
                agent[i] = divmod(int(a_cell), S)
                hold[i] = h
                ach[i] = rng.randint(0, 512)
                des[i] = rng.randint(1, 512) if rng.rand() < 0.7 else ach[i]
                stepn[i] = rng.randint(0, 10)
        env.set_state(grid=grids, init_grid=inits, agent_rc=agent, hold=hold, achieved=ach, desired=des, step_num=stepn)
        acts = torch.as_tensor(np.tile(np.arange(6), N // 6).astype(np.int32), device=env.device)
        obs, rew, done, info = env.step(acts)
        st = env.get_state()
        rew, done = rew.cpu().numpy(), done.cpu().numpy()
        frames = obs['observation'].cpu().numpy()
        o = OracleEnv(size=(S, S), max_steps=10, reward_style=style)
        for i in range(N):
            o.set_state(grids[i], inits[i], agent[i], hold[i], int(ach[i]), int(des[i]), int(stepn[i]))
            oo, r, d, _ = o.step(i % 6)
            s = o.state()
            tag = (style, i, 'action', i % 6)
            assert r == rew[i] and d == done[i], tag
            assert np.array_equal(st['grid'][i], s['grid']), tag
            assert tuple(st['agent_rc'][i]) == s['agent'] and st['hold'][i] == s['hold'], tag
            assert st['achieved'][i] == s['achieved'], (tag, bin(st['achieved'][i]), bin(s['achieved']))
            assert st['step_num'][i] == s['step_num'], tag
            assert np.array_equal(frames[i], s['obs']), tag
        env.close()


def test_invalid_actions_counted_not_executed():
    from gym_craftingworld_amd import CraftingWorldVecEnv
    env = CraftingWorldVecEnv(64, size=(5, 5), max_steps=9, obs_mode='state')
    env.reset()
    before = env.get_state()
    env.step(torch.full((64,), 6, dtype=torch.int64, device=env.device))
    env.step(torch.full((64,), -1, dtype=torch.int32, device=env.device))
    after = env.get_state()
    assert np.array_equal(before['grid'], after['grid']) and np.array_equal(before['agent_rc'], after['agent_rc'])
    assert (after['step_num'] == 2).all() and int(env.counters[3].item()) == 128
    assert (env.reward.cpu().numpy() == -1).all()
    env.close()


# ------------------------------------------------------------------ (4) full-size properties
def test_shard_equivalence_and_batch_position_invariance():
    """GPU g of G owns envs [g*N/G,(g+1)*N/G): running the shards separately equals the single batch."""
    from gym_craftingworld_amd import CraftingWorldVecEnv
    N, T, kw = 512, 80, dict(size=(8, 8), max_steps=30)
    keys, pos = _np_states(N, 4000)
    acts = torch.as_tensor(np.random.RandomState(3).randint(0, 6, size=(T, N)).astype(np.int32), device='cuda')

    def run(lo, hi):
        env = CraftingWorldVecEnv(hi - lo, obs_mode='pixels', **kw)
        env.set_rng_states(keys[lo:hi], pos[lo:hi])
        env.reset()
        rs, ds = [], []
        for t in range(T):
            o, r, d, _ = env.step(acts[t, lo:hi].contiguous())
            rs.append(r.clone())
            ds.append(d.clone())
        out = (torch.stack(rs).cpu().numpy(), torch.stack(ds).cpu().numpy(), o['observation'].cpu().numpy(),
               o['desired_goal'].cpu().numpy(), env.get_state())
        env.close()
        return out

    whole = run(0, N)
    parts = [run(0, 128), run(128, 320), run(320, N)]
    assert np.array_equal(whole[0], np.concatenate([p[0] for p in parts], axis=1))
    assert np.array_equal(whole[1], np.concatenate([p[1] for p in parts], axis=1))
    assert np.array_equal(whole[2], np.concatenate([p[2] for p in parts], axis=0))
    assert np.array_equal(whole[3], np.concatenate([p[3] for p in parts], axis=0))
    for k in whole[4]:
        assert np.array_equal(whole[4][k], np.concatenate([p[4][k] for p in parts], axis=0)), k


@pytest.mark.parametrize('size,max_steps', [(21, 300), (32, 40)])
def test_full_size_properties_65536(size, max_steps, monkeypatch):
    """BASELINE configs 3 and 5 at full batch size, through size-independent properties: the frame is a pure function of the state
    (full-frame render == dirty-cell repaint == render() of the current state), counters add up -- and a SAMPLE of the envs against the
    CPU oracle, all three frames: the first 64 envs, a stride through the batch, the LAST 32 (the array's partial last piece), and the envs
    on either side of every chunk boundary of the sweep (CW_TUNE_RENDER_CHUNK_ROUNDS makes it three launches: 24 576 + 24 576 + 16 384 envs)."""
    from gym_craftingworld_amd import CraftingWorldVecEnv
    from oracle import OracleBatch
    N, T = 65536, 45
    kw = dict(size=(size, size), max_steps=max_steps)
    monkeypatch.setenv('CW_TUNE_RENDER_CHUNK_ROUNDS', '150' if size == 21 else '350')
    full = CraftingWorldVecEnv(N, obs_mode='pixels', seed=123, **kw)
    monkeypatch.delenv('CW_TUNE_RENDER_CHUNK_ROUNDS')
    dirty = CraftingWorldVecEnv(N, obs_mode='pixels_dirty', seed=123, **kw)
    keys, pos = full.get_rng_states()
    idx = sorted(set(list(range(64)) + [24574, 24575, 24576, 24577, 49150, 49151, 49152, 49153] + list(range(1000, N, 4099))[:48] + list(range(N - 32, N))))
    idx_t = torch.as_tensor(idx, device='cuda')
    ora = OracleBatch(len(idx), rng_states=[(keys[i], int(pos[i])) for i in idx], **kw)
    full.reset()
    dirty.reset()
    ora.reset()
    gen = torch.Generator(device='cuda').manual_seed(9)
    dones = 0
    for t in range(T):
        a = torch.randint(0, 6, (N,), device='cuda', dtype=torch.int32, generator=gen)
        of, rf, df, _ = full.step(a)
        od, rd, dd, _ = dirty.step(a)
        assert torch.equal(rf, rd) and torch.equal(df, dd), t
        o_rew, o_done = ora.step(a[idx_t].cpu().numpy())
        assert np.array_equal(rf[idx_t].cpu().numpy(), o_rew) and np.array_equal(df[idx_t].cpu().numpy(), o_done), t
        dones += int(df.sum().item())
        if t % 11 == 0 or t == T - 1:
            assert torch.equal(of['observation'], od['observation']), t
            assert torch.equal(of['desired_goal'], od['desired_goal']), t
            assert torch.equal(of['init_observation'], od['init_observation']), t
            assert torch.equal(full.render(), of['observation']), t
    obs, des, ini = (of[k][idx_t].cpu().numpy() for k in ('observation', 'desired_goal', 'init_observation'))
    for j_, s in enumerate(ora.states()):
        assert np.array_equal(obs[j_], s['obs']) and np.array_equal(des[j_], s['desired_img']) and np.array_equal(ini[j_], s['init_img']), idx[j_]
    assert torch.equal(full.hdr, dirty.hdr) and torch.equal(full.slot_pos, dirty.slot_pos)
    assert int(full.counters[1].item()) == dones and int(full.counters[0].item()) == N * T
    if max_steps <= T:
        assert dones >= N        # every env timed out at least once (synchronized resets inside the window)
    full.close()
    dirty.close()


def test_single_env_facade_matches_golden():
    """craftingworld-v3 surface (N=1, numpy, no auto-reset) on BASELINE config 1's known-answer
    vector: RandomState(12345) reset + 300 actions from RandomState(999)."""
    import gym_craftingworld_amd as g
    env = g.make('craftingworld-v3')
    env.np_random = np.random.RandomState(12345)              # exactly what SURVEY 8c's reference session did
    assert env.np_random.get_state()[2] % 624 == 0            # (reading gives a snapshot of the device-resident stream)
    obs = env.reset()
    assert env.np_random.get_state()[2] == 617                # "MT pos after reset 617"
    assert obs['achieved_goal'] is obs['observation']
    assert env.desired_goal_vector.tolist() == [[1, 0, 0, 1, 0, 0, 0, 1, 0]] and env.agent_pos == (8, 5)
    assert crc(obs['observation']) == 0xf34f1edc and crc(obs['desired_goal']) == 0x9e9a73eb
    first = obs['observation']
    total, done = 0, False
    for a in np.random.RandomState(999).randint(0, 6, size=300):
        obs, r, done, info = env.step(a)
        total += r
    assert obs['observation'] is first                       # live alias, mutated in place (ray.py:359)
    assert total == -300 and done and env.agent_pos == (18, 5) and crc(obs['observation']) == 0x2c3d814f
    assert info['achieved_goal'].sum() == 0
    with pytest.raises(IndexError):
        env.step(6)
    env.close()
    flat = g.make('craftingworldflat-v3')
    o = flat.reset()
    assert o.shape == (32, 32, 3)
    o2, r, d, _ = flat.step(1)
    assert o2 is o
    flat.close()
    oh = g.make('craftingworldonehot-v3', size=(6, 6))
    o = oh.reset()
    assert o['observation'].shape == (6, 6, 12) and o['observation'][:, :, 8].sum() == 1
    assert o['desired_goal'][:, :, 8].sum() == 1
    oh.close()


@pytest.mark.parametrize('obs_mode', ['pixels', 'pixels_dirty', 'state'])
def test_step_captured_in_hip_graph(obs_mode):
    """One cw_step is a fixed launch sequence (no host-side state in kernel arguments), so it can be captured into a HIP graph and
    replayed: the replayed env must stay bit-identical to an eagerly stepped twin.  (A step captured on its own carries the look-ahead refill
    with it -- cw_step sees the capture -- so the replayed env refills on every step where the eager one does on every max_steps/4-th (8 ... 64): same results.)"""
    from gym_craftingworld_amd import CraftingWorldVecEnv
    N, T, kw = 2048, 90, dict(size=(7, 7), max_steps=25)
    keys, pos = _np_states(N, 99)
    envs = []
    for _ in range(2):
        e = CraftingWorldVecEnv(N, obs_mode=obs_mode, **kw)
        e.set_rng_states(keys, pos)
        e.reset()
        envs.append(e)
    eager, graphed = envs
    acts = torch.randint(0, 6, (T, N), device='cuda', dtype=torch.int32, generator=torch.Generator(device='cuda').manual_seed(4))
    static_a = torch.zeros(N, dtype=torch.int32, device='cuda')
    # step 0 eagerly on a side stream (torch's capture warm-up protocol), then capture one step
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        static_a.copy_(acts[0])
        graphed.step_async(static_a)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        graphed.step_async(static_a)
    # the capture itself did not execute: replay from step 1 on
    eager.step(acts[0])
    for t in range(1, T):
        static_a.copy_(acts[t])
        g.replay()
        o, r, d, _ = eager.step(acts[t])
        assert torch.equal(r, graphed.reward) and torch.equal(d, graphed.done), t
    assert torch.equal(eager.hdr, graphed.hdr) and torch.equal(eager.slot_pos, graphed.slot_pos)
    if obs_mode != 'state':
        for k in ('observation', 'desired_goal', 'init_observation'):
            assert torch.equal(eager._observation()[k], graphed._observation()[k]), k
    assert int(eager.counters[1].item()) == int(graphed.counters[1].item()) > 0
    for e in envs:
        e.close()


@pytest.mark.gpu
@pytest.mark.parametrize('obs_mode,K', [('state', 5), ('state', 48), ('pixels_dirty', 7), ('pixels', 20)])
def test_step_many_and_captured_graphs_equal_stepping(obs_mode, K):
    """VecEnv.step_many (cw_step_many: K steps from an action array in one library call) and VecEnv.capture_steps (the same K steps as a HIP
    graph that re-reads the action ring on every replay) against an eagerly stepped twin: rewards and dones of every K-th step, and at the end
    state, frames, counters and random streams.  Episodes of 25 steps: finished envs take look-ahead records and the refill kernel rides
    inside the sequences (K = 5 < the refill period: a captured graph carries its own)."""
    from gym_craftingworld_amd import CraftingWorldVecEnv
    N, rounds, kw = 3000, 8, dict(size=(7, 7), max_steps=25, seed=5)
    envs = [CraftingWorldVecEnv(N, obs_mode=obs_mode, **kw) for _ in range(3)]
    for e in envs:
        e.reset()
    eager, many, graphed = envs
    gen = torch.Generator(device='cuda').manual_seed(8)
    ring = torch.zeros((K, N), dtype=torch.uint8, device='cuda')
    graph = graphed.capture_steps(ring)                      # (capturing takes no step: the twins start level)
    assert int(graphed.counters[0]) == 0
    for r_ in range(rounds):
        acts = torch.randint(0, 6, (K, N), device='cuda', dtype=torch.uint8, generator=gen)
        ring.copy_(acts)
        graph.replay()
        many.step_many(acts)
        for t in range(K):
            eager.step(acts[t])
        for e in (many, graphed):
            assert torch.equal(e.reward, eager.reward) and torch.equal(e.done, eager.done), (r_, e is many)
    for e in (many, graphed):
        assert torch.equal(e.hdr, eager.hdr) and torch.equal(e.slot_pos, eager.slot_pos) and torch.equal(e.counters, eager.counters)
        if obs_mode != 'state':
            for k in ('observation', 'desired_goal', 'init_observation'):
                assert torch.equal(e._observation()[k], eager._observation()[k]), k
    assert int(eager.counters[1].item()) > N
    ke, pe = eager.get_rng_states()
    for e in (many, graphed):
        k2, p2 = e.get_rng_states()
        assert np.array_equal(k2, ke) and np.array_equal(p2, pe)
    for e in envs:
        e.close()


@pytest.mark.parametrize('obs_mode', ['pixels', 'pixels_dirty'])
def test_terminal_observation_vs_oracle(obs_mode):
    """keep_terminal_obs: where done, info['terminal_observation'] is the oracle's frame BEFORE its
    reset, and obs is the first frame of the next episode (gym.vector auto-reset semantics)."""
    from gym_craftingworld_amd import CraftingWorldVecEnv
    from oracle import OracleBatch
    N, T, kw = 160, 70, dict(size=(5, 5), max_steps=12)
    keys, pos = _np_states(N, 555)
    env = CraftingWorldVecEnv(N, obs_mode=obs_mode, keep_terminal_obs=True, **kw)
    env.set_rng_states(keys, pos)
    ora = OracleBatch(N, rng_states=list(zip(keys, pos)), **kw)
    env.reset()
    ora.reset()
    acts = np.random.RandomState(8).randint(0, 6, size=(T, N)).astype(np.int32)
    dacts = torch.as_tensor(acts, device=env.device)
    seen = 0
    for t in range(T):
        obs, rew, done, info = env.step(dacts[t])
        term = info['terminal_observation'].cpu().numpy()
        cur = obs['observation'].cpu().numpy()
        d = done.cpu().numpy()
        for i, e in enumerate(ora.envs):
            o, r, dd, _ = e.step(int(acts[t, i]))
            assert dd == d[i]
            if dd:
                assert np.array_equal(term[i], o['observation']), (t, i)
                e.reset()
                seen += 1
            assert np.array_equal(cur[i], e.state()['obs']), (t, i)
    assert seen > N
    env.close()


@pytest.mark.parametrize('obs_mode', ['pixels_dirty', 'pixels'])
def test_multi_device_facade_and_gymnasium_adapter(obs_mode):
    """One process driving several engines (here: two engines, each on a stream of its own, on the one visible GPU) equals the
    single batch, in the dirty-cell and the full-frame mode; the gymnasium adaptor splits done into terminated (success) / truncated (time-out)."""
    from gym_craftingworld_amd import CraftingWorldVecEnv
    from gym_craftingworld_amd.adapters import GymnasiumVecAdapter, MultiDeviceVecEnv
    N, T, kw = 300, 60, dict(size=(5, 5), max_steps=15, obs_mode=obs_mode)
    keys, pos = _np_states(N, 2222)
    acts = torch.as_tensor(np.random.RandomState(1).randint(0, 6, size=(T, N)).astype(np.int32), device='cuda')
    single = CraftingWorldVecEnv(N, **kw)
    single.set_rng_states(keys, pos)
    multi = MultiDeviceVecEnv(N, ['cuda:0', 'cuda:0'], **kw)
    multi.set_rng_states(keys, pos)
    assert multi.ranges == [(0, 150), (150, 300)]
    gs = GymnasiumVecAdapter(single)
    obs, info = gs.reset()
    multi.reset()
    n_term = n_trunc = 0
    for t in range(T):
        o, r, term, trunc, info = gs.step(acts[t])
        outs = multi.step(acts[t])
        multi.synchronize()
        assert torch.equal(r, torch.cat([x[1] for x in outs]))
        assert torch.equal(term | trunc, torch.cat([x[2] for x in outs]))
        assert torch.equal(o['observation'], torch.cat([x[0]['observation'] for x in outs]))
        assert not bool((term & trunc).any())
        assert bool((r[term] == 15).all()) and bool((r[trunc] == -1).all())
        # gymnasium.vector's episode statistics: info['episode'] = {'r': return, 'l': length} with the mask info['_episode'] -- a success at step l
        # returns 15 - (l - 1), a time-out -15 (ray.py:361-367 summed by the loop)
        ep, fin = info['episode'], info['_episode']
        assert torch.equal(fin, term | trunc)
        assert bool((ep['r'][term] == 16 - ep['l'][term]).all()) and bool((ep['r'][trunc] == -15).all()) and bool((ep['l'][trunc] == 15).all())
        n_term += int(term.sum().item())
        n_trunc += int(trunc.sum().item())
    assert n_trunc > 0 and n_term > 0
    gs.close()
    multi.close()


@pytest.mark.parametrize('case', [dict(N=4096, T=140, kw=dict(size=(21, 21), max_steps=50)),
                                  dict(N=1000, T=90, kw=dict(size=(5, 5), max_steps=12, reward_style='subset')),
                                  dict(N=130, T=70, kw=dict(size=(6, 6), max_steps=10, fixed_init_state=3))],
                         ids=lambda c: 'N%d_S%d' % (c['N'], c['kw']['size'][0]))
def test_persistent_rollout_equals_stepping(case):
    """cw_rollout (T steps in one persistent kernel) == T x cw_step: rewards and dones of every step,
    final state, goal state, RNG streams and counters; and both equal the oracle on the first envs."""
    from gym_craftingworld_amd import CraftingWorldVecEnv
    from gym_craftingworld_amd import _lib as L
    from oracle import OracleBatch
    N, T, kw = case['N'], case['T'], case['kw']
    keys, pos = _np_states(N, 8080)
    envs = []
    for _ in range(2):
        e = CraftingWorldVecEnv(N, obs_mode='state', **kw)
        e.set_rng_states(keys, pos)
        if kw.get('fixed_init_state'):
            L.check(e._lib.cw_generate_fixed_states(e._h, e._stream()), 'pool')
        e.reset()
        envs.append(e)
    stepped, rolled = envs
    acts = torch.randint(0, 6, (T, N), device='cuda', dtype=torch.uint8, generator=torch.Generator(device='cuda').manual_seed(21))
    rew, don = rolled.rollout(acts)
    rs, ds = [], []
    for t in range(T):
        _, r, d, _ = stepped.step(acts[t])
        rs.append(r.clone())
        ds.append(d.clone())
    assert torch.equal(rew, torch.stack(rs)) and torch.equal(don, torch.stack(ds))
    a, b = stepped.get_state(), rolled.get_state()
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    assert torch.equal(stepped.counters, rolled.counters)
    assert torch.equal(stepped.reward, rolled.reward) and torch.equal(stepped.done, rolled.done)
    assert torch.equal(stepped.achieved_mask, rolled.achieved_mask)
    ka, pa = stepped.get_rng_states()
    kb, pb = rolled.get_rng_states()
    assert np.array_equal(pa, pb) and np.array_equal(ka[:, 1:], kb[:, 1:])
    M = min(N, 96)
    ora = OracleBatch(M, rng_states=[(keys[i], int(pos[i])) for i in range(M)], **kw)
    ora.reset()
    _, o_rew, o_don = ora.rollout(acts[:, :M].cpu().numpy().astype(np.int8), nthreads=4, record=True)
    assert np.array_equal(rew[:, :M].cpu().numpy(), o_rew) and np.array_equal(don[:, :M].cpu().numpy(), o_don.astype(bool))
    for e in envs:
        e.close()


@pytest.mark.parametrize('size,n_envs', [(21, 12288), (32, 4096), (9, 8192)])
def test_reset_soak_vs_oracle(size, n_envs):
    """Reset-heavy soak: max_steps=2 forces a reset of every env every other step, so the
    lane-parallel rejection sampling, the token bookkeeping and imagine_obs are compared with the
    oracle on >100 000 independent MT19937 streams/resets (placement, desired mask, goal state, agent,
    final stream position)."""
    from gym_craftingworld_amd import CraftingWorldVecEnv
    from oracle import OracleBatch
    T = 16
    kw = dict(size=(size, size), max_steps=2)
    env = CraftingWorldVecEnv(n_envs, obs_mode='state', seed=77, **kw)
    keys, pos = env.get_rng_states()
    ora = OracleBatch(n_envs, rng_states=[(keys[i], int(pos[i])) for i in range(n_envs)], **kw)
    env.reset()
    ora.reset()
    acts = np.random.RandomState(3).randint(0, 6, size=(T, n_envs)).astype(np.int8)
    env.rollout(torch.as_tensor(acts.astype(np.uint8), device=env.device), record=False)   # persistent kernel: same reset code
    total = ora.rollout(acts, nthreads=8)
    assert total == T * n_envs
    st = env.get_state()
    k2, p2 = env.get_rng_states()
    for i, s in enumerate(ora.states()):
        assert np.array_equal(st['grid'][i], s['grid']), i
        assert np.array_equal(st['init_grid'][i], s['init_grid']), i
        assert np.array_equal(st['goal_grid'][i], s['goal_grid']), i
        assert tuple(st['agent_rc'][i]) == s['agent'] and tuple(st['goal_agent_rc'][i]) == s['goal_agent'], i
        assert st['desired'][i] == s['desired'] and st['ep_no'][i] == s['ep_no'], i
    for i in range(0, n_envs, 97):
        ok, op = ora.envs[i].get_rng()
        assert p2[i] % 624 == op % 624, i
        if op % 624:
            assert np.array_equal(k2[i][1:], ok[1:]), i
    assert int(env.counters[1].item()) == n_envs * (T // 2)
    env.close()


def test_config4_shard_shape_mixed_menus_131072():
    """BASELINE configs[3] as one rank sees it: 131 072 envs per GPU (1M over 8), env i using ordered
    task menu i mod 8, full-frame pixels.  A strided sample of envs is replayed by the oracle."""
    from gym_craftingworld_amd import CraftingWorldVecEnv
    from oracle import OracleBatch
    N, T = 131072, 24
    menus = [dict(selected_tasks=TASKS), dict(selected_tasks=TASKS[:3]), dict(selected_tasks=TASKS[3:], number_of_tasks=2),
             dict(selected_tasks=['GoToHouse']), dict(selected_tasks=TASKS[::-1], stacking=False),
             dict(selected_tasks=['MoveAxe', 'MoveHammer', 'MoveSticks'], reward_style='subset'),
             dict(selected_tasks=['EatBread', 'MakeBread'], number_of_tasks=1), dict(selected_tasks=TASKS[1::2])]
    env_menu = (np.arange(N) % len(menus)).astype(np.uint8)
    kw = dict(size=(21, 21), max_steps=10)
    env = CraftingWorldVecEnv(N, obs_mode='pixels', seed=4242, task_menus=menus, env_menu=env_menu, **kw)
    keys, pos = env.get_rng_states()
    sample = np.unique(np.concatenate([np.arange(0, N, 1021)[:120], np.arange(N - 8, N)]))      # (a stride through the batch and the last envs: the array's last piece)
    ora = OracleBatch(len(sample), rng_states=[(keys[i], int(pos[i])) for i in sample],
                      per_env_kwargs=[menus[env_menu[i]] for i in sample], **kw)
    env.reset()
    ora.reset()
    gen = torch.Generator(device='cuda').manual_seed(1)
    sidx = torch.as_tensor(sample, device='cuda')
    for t in range(T):
        a = torch.randint(0, 6, (N,), device='cuda', dtype=torch.int32, generator=gen)
        obs, rew, done, _ = env.step(a)
        o_rew, o_done = ora.step(a[sidx].cpu().numpy())
        assert np.array_equal(rew[sidx].cpu().numpy(), o_rew) and np.array_equal(done[sidx].cpu().numpy(), o_done), t
    frames = obs['observation'][sidx].cpu().numpy()
    goals = obs['desired_goal'][sidx].cpu().numpy()
    for j, s in enumerate(ora.states()):
        assert np.array_equal(frames[j], s['obs']) and np.array_equal(goals[j], s['desired_img']), sample[j]
    inits = obs['init_observation'][sidx].cpu().numpy()
    for j, s in enumerate(ora.states()):
        assert np.array_equal(inits[j], s['init_img']), sample[j]
    assert int(env.counters[0].item()) == N * T and int(env.counters[1].item()) >= 2 * N
    env.close()


@pytest.mark.parametrize('obs_mode,S,N', [('pixels', 6, 200), ('pixels_dirty', 6, 200), ('pixels', 13, 151), ('pixels', 21, 67)])
def test_altobs_raster_vs_oracle(obs_mode, S, N):
    """the AltObs rasteriser (3x3-px CPV tiles + holding strip) in both pixel modes
    vs the oracle: all three frames, cw_render, terminal frames; includes sticks held over sticks (2 x colour).
    (All sizes go through the sweep of aligned 4-KiB pieces: a piece of the 6x6 array overlaps up to five 1 134-byte frames, of the 13x13 and
    21x21 arrays two; the arrays end in a partial piece.)"""
    from gym_craftingworld_amd import CraftingWorldVecEnv
    from oracle import OracleBatch
    T, kw = 90, dict(size=(S, S), max_steps=30)
    keys, pos = _np_states(N, 6060)
    env = CraftingWorldVecEnv(N, obs_mode=obs_mode, raster='alt', keep_terminal_obs=True, **kw)
    env.set_rng_states(keys, pos)
    ora = OracleBatch(N, rng_states=list(zip(keys, pos)), alt_obs=True, **kw)
    obs = env.reset()
    ora.reset()
    assert tuple(obs['observation'].shape) == (N, 3 * S + 3, 3 * S, 3)
    assert env.render_kernel_name() == ('cw_step_fused_kernel' if obs_mode == 'pixels_dirty' else 'cw_render_pieces_kernel')      # (AltObs raster: never fused)
    acts = np.random.RandomState(2).randint(0, 6, size=(T, N)).astype(np.int32)
    dacts = torch.as_tensor(acts, device=env.device)
    for t in range(T):
        obs, rew, done, info = env.step(dacts[t])
        term = info['terminal_observation'].cpu().numpy() if (t % 9 == 0) else None
        d = done.cpu().numpy()
        for i, e in enumerate(ora.envs):
            o, r, dd, _ = e.step(int(acts[t, i]))
            assert dd == d[i] and r == int(rew[i].item()) if t % 9 == 0 else dd == d[i]
            if dd:
                if term is not None:
                    assert np.array_equal(term[i], o['observation']), (t, i)
                e.reset()
        if t % 9 == 0 or t == T - 1:
            cur, des, ini, ren = (x.cpu().numpy() for x in (obs['observation'], obs['desired_goal'], obs['init_observation'], env.render()))
            for i, s in enumerate(ora.states()):
                assert np.array_equal(cur[i], s['obs']), (t, i)
                assert np.array_equal(ren[i], s['obs']), (t, i)
                assert np.array_equal(des[i], s['desired_img']) and np.array_equal(ini[i], s['init_img']), (t, i)
    # the double-count corner: sticks in hand, standing on sticks -> blue channel 2*160 mod 256 = 64
    grid = np.zeros((N, S, S), np.uint8)
    grid[:, 2, 3] = 1
    env.set_state(grid=grid, agent_rc=np.tile(np.array([[2, 3]], np.uint8), (N, 1)), hold=np.ones(N, np.uint8))
    fr = env.render()[0].cpu().numpy()
    assert fr[6, 9].tolist() == [90, 164, 64] and fr[8, 11].tolist() == [0, 0, 255] and fr[3 * S:, 3:6].min() == 255
    env.close()


@pytest.mark.parametrize('raster,shift', [('ray', 4), ('ray', 12), ('alt', 1), ('alt', 6)])
def test_render_into_a_buffer_of_any_alignment(raster, shift):
    """cw_render paints a caller-supplied array: a 16-byte aligned one with the sweep of aligned pieces, any other pointer the rasteriser's
    stores accept with the older painters -- the same frames either way, and nothing outside them."""
    from gym_craftingworld_amd import CraftingWorldVecEnv
    N = 777
    env = CraftingWorldVecEnv(N, size=(21, 21), max_steps=50, obs_mode='pixels', raster=raster, seed=3)
    env.reset()
    gen = torch.Generator(device='cuda').manual_seed(8)
    for t in range(12):
        env.step(torch.randint(0, 6, (N,), device='cuda', dtype=torch.uint8, generator=gen))
    want = env.render()
    assert want.data_ptr() % 16 == 0 and torch.equal(want, env._obs)
    n = want.numel()
    raw = torch.full((n + 64,), 7, dtype=torch.uint8, device='cuda')
    view = raw[shift:shift + n].view(want.shape)
    assert view.data_ptr() % 16 == shift
    got = env.render(out=view)
    assert torch.equal(got, want)
    assert int(raw[:shift].min()) == 7 and int(raw[shift + n:].min()) == 7 and int(raw[shift + n:].max()) == 7
    env.close()


def test_compute_reward_batch_matches_reference_rules():
    from gym_craftingworld_amd import CraftingWorldVecEnv
    env = CraftingWorldVecEnv(4, size=(5, 5), max_steps=77, obs_mode='state')
    a = torch.arange(512, device='cuda').repeat_interleave(512)
    d = torch.arange(512, device='cuda').repeat(512)
    for subset in (False, True):
        got = env.compute_reward_batch(a, d, subset=subset).cpu().numpy()
        av = ((a.cpu().numpy()[:, None] >> np.arange(9)) & 1).astype(np.int64)
        dv = ((d.cpu().numpy()[:, None] >> np.arange(9)) & 1).astype(np.int64)
        if subset:
            ref = np.where((dv - av).max(axis=1) == 0, 77, -1)           # ray.py:763-767
        else:
            ref = np.where((av == dv).all(axis=1), 77, -1)               # ray.py:757-761
        assert np.array_equal(got, ref)
    env.close()


def test_overlap_stress_desynchronised_resets():
    """The full-pixel step forks: main stream paints non-done envs while the side stream resets done
    envs and paints their three frames.  Short, de-synchronised episodes (max_steps=23, random phases
    from successes) keep both streams busy on every step; full-frame and dirty-cell engines must stay
    identical throughout, and the first envs equal to the oracle."""
    from gym_craftingworld_amd import CraftingWorldVecEnv
    from oracle import OracleBatch
    N, T, M = 32768, 700, 96
    kw = dict(size=(21, 21), max_steps=23)
    full = CraftingWorldVecEnv(N, obs_mode='pixels', seed=31, keep_terminal_obs=True, **kw)
    dirty = CraftingWorldVecEnv(N, obs_mode='pixels_dirty', seed=31, keep_terminal_obs=True, **kw)
    keys, pos = full.get_rng_states()
    # scatter the episode phases so that some envs finish on every step
    phase = (np.arange(N) * 7 % 23).astype(np.int32)
    ora = OracleBatch(M, rng_states=[(keys[i], int(pos[i])) for i in range(M)], **kw)
    full.reset(); dirty.reset(); ora.reset()
    full.set_state(step_num=phase); dirty.set_state(step_num=phase)
    for i, e in enumerate(ora.envs):
        s = e.state()
        e.set_state(s['grid'], s['init_grid'], s['agent'], s['hold'], s['achieved'], s['desired'], int(phase[i]))
    gen = torch.Generator(device='cuda').manual_seed(5)
    for t in range(T):
        a = torch.randint(0, 6, (N,), device='cuda', dtype=torch.uint8, generator=gen)
        of, rf, df, inf_f = full.step(a)
        od, rd, dd, inf_d = dirty.step(a)
        if t % 50 == 0 or t == T - 1:
            assert torch.equal(rf, rd) and torch.equal(df, dd), t
            for k in ('observation', 'desired_goal', 'init_observation'):
                assert torch.equal(of[k], od[k]), (t, k)
            m = df.nonzero().squeeze(1)
            assert m.numel() > 0
            assert torch.equal(inf_f['terminal_observation'][m], inf_d['terminal_observation'][m]), t
        o_rew, o_done = ora.step(a[:M].cpu().numpy())
        assert np.array_equal(rf[:M].cpu().numpy(), o_rew) and np.array_equal(df[:M].cpu().numpy(), o_done), t
    for i, s in enumerate(ora.states()):
        assert np.array_equal(of['observation'][i].cpu().numpy(), s['obs']), i
        assert np.array_equal(of['desired_goal'][i].cpu().numpy(), s['desired_img']), i
    assert torch.equal(full.hdr, dirty.hdr) and torch.equal(full.counters, dirty.counters)
    full.close(); dirty.close()


def test_episode_recorder_writes_gifs(tmp_path):
    from PIL import Image
    from gym_craftingworld_amd import CraftingWorldVecEnv
    from gym_craftingworld_amd.recorder import EpisodeRecorder
    env = CraftingWorldVecEnv(8, size=(5, 5), max_steps=6, obs_mode='pixels', keep_terminal_obs=True, seed=3)
    rec = EpisodeRecorder(env, env_index=2, out_dir=str(tmp_path), scale=2)
    rec.after_reset(env.reset())
    paths = []
    gen = torch.Generator(device='cuda').manual_seed(11)
    for t in range(20):
        obs, rew, done, info = env.step(torch.randint(0, 6, (8,), device='cuda', generator=gen))
        p = rec.after_step(obs, done, info)
        if p:
            paths.append(p)
    assert len(paths) >= 3
    im = Image.open(paths[0])
    # reset frame + 6 steps (pillow merges identical consecutive frames); obs | desired_goal side by side
    assert 2 <= im.n_frames <= 7 and im.size == (2 * 2 * 20, 2 * 20)
    env.close()


def test_process_exit_with_live_views_is_clean():
    """A process that exits without close() while torch views of engine buffers are alive must end
    normally (engines are destroyed at exit before the HIP runtime, DLPack deleters are C functions)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rc = subprocess.call([sys.executable, os.path.join(root, 'tools', 'microbench', 'exit_probe.py')], cwd=root,
                         stderr=subprocess.DEVNULL)
    assert rc == 3


def test_error_behaviour_of_the_boundary():
    """Status codes of the C ABI surface as Python exceptions: CW_ERR_INVALID -> ValueError, CW_ERR_STATE /
    CW_ERR_HIP -> CraftingWorldError; reference-style errors for bad task names and action indices."""
    import ctypes as C
    import gym_craftingworld_amd as g
    from gym_craftingworld_amd import _lib as L
    with pytest.raises(ValueError):
        g.CraftingWorldVecEnv(4, size=(3, 3))                                   # < 12 cells: sample_state cannot place 9 items
    with pytest.raises(ValueError):
        g.CraftingWorldVecEnv(4, max_steps=70000)
    with pytest.raises(ValueError):
        g.CraftingWorldVecEnv(4, selected_tasks=['MakeBread', 'Fly'])           # list.index raises ValueError, ray.py:174
    with pytest.raises(ValueError):
        g.CraftingWorldVecEnv(4, task_list=TASKS[:5], selected_tasks=TASKS[:2])  # eval_task_edit needs all 9 slots
    with pytest.raises(ValueError):
        g.CraftingWorldVecEnv(4, obs_mode='rgb')
    with pytest.raises(NotImplementedError):
        g.CraftingWorldVecEnv(4, store_gif=True)
    env = g.CraftingWorldVecEnv(4, size=(5, 5), obs_mode='state')
    with pytest.raises(g.CraftingWorldError):
        env.step(torch.zeros(4, dtype=torch.int32, device='cuda'))              # step before reset
    env.reset()
    with pytest.raises(ValueError):
        env.step(torch.zeros(5, dtype=torch.int32, device='cuda'))
    with pytest.raises(ValueError):
        env.set_state(grid=np.full((4, 5, 5), 1, np.uint8))                     # 25 objects: not a reachable state
    with pytest.raises(ValueError):
        env.rollout(torch.zeros((3, 7), dtype=torch.uint8, device='cuda'))
    lib = L.load()
    assert lib.cw_step(None, None, 0, None) == L.CW_ERR_INVALID and b'null' in lib.cw_last_error()
    pix = g.CraftingWorldVecEnv(4, size=(5, 5), obs_mode='pixels')
    pix.reset()
    with pytest.raises(ValueError):
        pix.rollout(torch.zeros((3, 4), dtype=torch.uint8, device='cuda'))      # rollout is state-only
    one = g.make('craftingworld-v3', size=(5, 5))
    one.reset()
    for bad in (6, -7):
        with pytest.raises(IndexError):
            one.step(bad)                                                       # ACTIONS[action], ray.py:308
    assert one.step(-1)[1] == -1                                                # (-1 is a list index too: 'drop')
    for e in (env, pix, one):
        e.close()


def test_systematic_transition_table():
    """Every combination of (object in the target cell 0..8) x (object under the agent) x (hold) x (action)
    x (where the init grid put sticks/axe/hammer/tree relative to the two cells) x (achieved bits) x
    (agent interior / at the wall), one step each, HIP vs oracle -- the whole local transition table of
    step(), eval_task_edit() and both reward rules, not a random sample of it."""
    from itertools import product
    from gym_craftingworld_amd import CraftingWorldVecEnv
    from oracle import OracleEnv
    S = 5
    DR = [(-1, 0), (0, 1), (1, 0), (0, -1)]
    cases = []
    for tgt, under, hold, a, initv, achv, wall in product(range(9), (0, 1, 2, 3, 7, 8), range(4), range(6), range(4), range(3), (0, 1)):
        if hold and under and sum(1 for x in (tgt, under) if x) + 1 > 8:
            continue
        ar, ac = (0, 0) if wall else (2, 2)
        g = np.zeros((S, S), np.uint8)
        g[ar, ac] = under
        d = DR[a] if a < 4 else DR[1]
        tr, tc = ar + d[0], ac + d[1]
        if 0 <= tr < S and 0 <= tc < S:
            g[tr, tc] = tgt
        else:
            tr, tc = ar, ac                      # clamped move: target == own cell
        ig = np.zeros((S, S), np.uint8)          # one of each object; objects 1,2,3,5 placed per initv
        spots = {0: [(4, 0), (4, 1), (4, 2), (4, 3)],                       # all elsewhere
                 1: [(tr, tc), (4, 1), (4, 2), (4, 3)],                     # sticks started in the target cell
                 2: [(4, 0), (tr, tc), (4, 2), (4, 3)] if hold != 3 else [(4, 0), (4, 1), (tr, tc), (4, 3)],   # held tool's origin
                 3: [(4, 0), (4, 1), (4, 2), (tr, tc)]}[initv]              # tree started in the target cell
        for code, (r, c) in zip((1, 2, 3, 5), spots):
            ig[r, c] = code
        for code, (r, c) in zip((4, 6, 7, 8), [(3, 4), (2, 4), (1, 4), (0, 4)]):
            ig[r, c] = code
        ach = (0, 1 << 3, 0x1FF)[achv]
        cases.append((g, ig, (ar, ac), hold, a, ach))
    N = len(cases)
    for style in (None, 'subset'):
        env = CraftingWorldVecEnv(N, size=(S, S), max_steps=9, obs_mode='pixels_dirty', reward_style=style, auto_reset=False)
        env.reset()
        rng = np.random.RandomState(1)
        des = np.array([c[5] if rng.rand() < 0.3 else rng.randint(1, 512) for c in cases], np.uint16)
        env.set_state(grid=np.stack([c[0] for c in cases]), init_grid=np.stack([c[1] for c in cases]),
                      agent_rc=np.array([c[2] for c in cases], np.uint8), hold=np.array([c[3] for c in cases], np.uint8),
                      achieved=np.array([c[5] for c in cases], np.uint16), desired=des, step_num=np.full(N, 3, np.int32))
        obs, rew, done, _ = env.step(torch.as_tensor(np.array([c[4] for c in cases], np.int32), device=env.device))
        st = env.get_state()
        rew, done, frames = rew.cpu().numpy(), done.cpu().numpy(), obs['observation'].cpu().numpy()
        o = OracleEnv(size=(S, S), max_steps=9, reward_style=style)
        for i, (g, ig, ag, hold, a, ach) in enumerate(cases):
            o.set_state(g, ig, ag, hold, ach, int(des[i]), 3)
            _, r, d, _ = o.step(a)
            s = o.state()
            tag = (style, i, 'tgt/under/hold/a', int(g.max()), hold, a)
            assert (r, d) == (rew[i], done[i]), tag
            assert np.array_equal(st['grid'][i], s['grid']) and tuple(st['agent_rc'][i]) == s['agent'], tag
            assert st['hold'][i] == s['hold'] and st['achieved'][i] == s['achieved'], (tag, bin(st['achieved'][i]), bin(s['achieved']))
            assert np.array_equal(frames[i], s['obs']), tag
        env.close()
    assert N > 20000


@pytest.mark.gpu
@pytest.mark.parametrize('obs_mode,raster,N', [('state', 'ray', 1003), ('state', 'ray', 70001), ('pixels_dirty', 'ray', 5000),
                                               ('pixels_dirty', 'alt', 777), ('pixels', 'ray', 5000), ('pixels', 'alt', 777)])
def test_lookahead_records_equal_the_slow_path(obs_mode, raster, N, monkeypatch):
    """Engines that reset by themselves keep the outcome of every env's NEXT reset() ready (cw_refill_kernel, every max_steps/4 steps, 8 ... 64) and a finished
    env takes it over inside the step kernel; an env that finishes again before the next refill is reset the slow way, on the spot.
    CW_TUNE_LOOKAHEAD=0 keeps no records at all.  Same seeds and actions, episodes of at most 17 steps ending on every step (so both ways are
    taken all the time): every buffer, counter and RNG stream must agree, with batch sizes that leave the last wavefront partly filled."""
    from gym_craftingworld_amd import CraftingWorldVecEnv
    kw = dict(size=(9, 9), max_steps=17, obs_mode=obs_mode, raster=raster, seed=77)
    if obs_mode != 'state':
        kw['keep_terminal_obs'] = True
    one = CraftingWorldVecEnv(N, **kw)
    monkeypatch.setenv('CW_TUNE_LOOKAHEAD', '0')
    two = CraftingWorldVecEnv(N, **kw)
    monkeypatch.delenv('CW_TUNE_LOOKAHEAD')
    one.reset(); two.reset()
    phase = (np.arange(N) * 5 % 17).astype(np.int32)          # some envs finish on every step
    one.set_state(step_num=phase); two.set_state(step_num=phase)
    gen = torch.Generator(device='cuda').manual_seed(9)
    ended = 0
    for t in range(120):
        a = torch.randint(0, 6, (N,), device='cuda', dtype=torch.int64 if t % 2 else torch.uint8, generator=gen)
        o1, r1, d1, i1 = one.step(a)
        o2, r2, d2, i2 = two.step(a)
        assert torch.equal(r1, r2) and torch.equal(d1, d2), t
        ended += int(d1.sum())
        if t % 10 == 0 or t == 119:
            assert torch.equal(one.hdr, two.hdr) and torch.equal(one.slot_pos, two.slot_pos), t
            for k in ('achieved_goal', 'desired_goal'):
                assert torch.equal(i1[k], i2[k]), (t, k)
            m = d1.nonzero().squeeze(1)
            assert torch.equal(i1['episode_length'][m], i2['episode_length'][m]), t
            if obs_mode != 'state':
                for k in ('observation', 'desired_goal', 'init_observation'):
                    assert torch.equal(o1[k], o2[k]), (t, k)
                assert torch.equal(i1['terminal_observation'][m], i2['terminal_observation'][m]), t
    assert ended > N
    assert torch.equal(one.counters, two.counters)
    s1, s2 = one.get_state(), two.get_state()
    for k in s1:
        assert np.array_equal(s1[k], s2[k]), k
    k1, p1 = one.get_rng_states(); k2, p2 = two.get_rng_states()
    assert np.array_equal(k1, k2) and np.array_equal(p1, p2)
    one.close(); two.close()


@pytest.mark.gpu
@pytest.mark.parametrize('obs_mode', ['pixels_dirty', 'pixels', 'state'])
def test_host_mapped_outputs_equal_device_outputs(obs_mode):
    """cw_config.host_outputs (the single-env loop's engine): kernels write frames, reward, done and masks straight
    into pinned host memory; step()/reset() return CPU tensors after one stream sync and take host actions.
    Must equal an ordinary device-resident engine on the same seeds and actions, auto-reset included."""
    from gym_craftingworld_amd import CraftingWorldVecEnv
    N = 48
    kw = dict(size=(8, 8), max_steps=21, obs_mode=obs_mode, seed=5)
    if obs_mode != 'state':
        kw['keep_terminal_obs'] = True
    dev = CraftingWorldVecEnv(N, **kw)
    host = CraftingWorldVecEnv(N, host_outputs=True, **kw)
    od, oh = dev.reset(), host.reset()
    assert not host.reward.is_cuda and host.hdr.is_cuda
    if obs_mode != 'state':
        for k in od:
            assert not oh[k].is_cuda and torch.equal(od[k].cpu(), oh[k]), k
    assert torch.equal(dev.desired_mask.cpu(), host.desired_mask) and int(host.achieved_mask.abs().sum()) == 0
    rng = np.random.RandomState(3)
    ended = 0
    for t in range(150):
        a = rng.randint(0, 6, size=N)
        od, rd, dd, idv = dev.step(torch.as_tensor(a, device='cuda'))
        oh, rh, dh, ih = host.step(a if t % 2 else a.tolist())       # host actions: arrays or plain lists
        assert torch.equal(rd.cpu(), rh) and torch.equal(dd.cpu(), dh), t
        for k in ('achieved_goal', 'desired_goal', 'episode_length'):
            assert torch.equal(idv[k].cpu(), ih[k]), (t, k)
        if obs_mode != 'state':
            for k in od:
                assert torch.equal(od[k].cpu(), oh[k]), (t, k)
            m = dh.nonzero().squeeze(1)
            assert torch.equal(idv['terminal_observation'].cpu()[m], ih['terminal_observation'][m]), t
        ended += int(dh.sum())
    assert ended > N
    assert torch.equal(dev.hdr, host.hdr) and torch.equal(dev.counters, host.counters)
    # a device tensor of actions is still accepted
    a = torch.randint(0, 6, (N,), device='cuda')
    _, rd, _, _ = dev.step(a)
    _, rh, _, _ = host.step(a)
    assert torch.equal(rd.cpu(), rh)
    dev.close(); host.close()


@pytest.mark.gpu
@pytest.mark.parametrize('name,variant', [('ray5_scripted', 'dict'), ('ray8_selected', 'int64'), ('ray5_subset', 'flat'),
                                          ('ray8_scripted', 'onehot'), ('alt5_random', 'alt'), ('alt8_scripted', 'alt_stacked'),
                                          ('alt4_double', 'alt_exact'), ('alt8_scripted', 'alt_exact'), ('alt4_double', 'alt_exact_stacked')])
def test_single_env_facades_replay_fixtures(name, variant):
    """The gym.Env-shaped N=1 classes (host-mapped outputs, no auto-reset, numpy in/out) on whole reference
    trajectories: every reward, done, achieved mask, frame and reset of the fixture, through each façade's own
    return convention (Dict / int64 copies / Flat's bare frame / OneHot states / AltObs Dict and stacked).  alt_exact: the AltObs class
    with reference_dtypes=True against the fixtures' int16 CRCs -- the reference's int image EXACTLY, values above 255 included
    (alt4_double holds sticks over sticks on 9 steps: (90, 164, 320), craftingworld_altobs.py:527-543)."""
    import gym_craftingworld_amd as cw
    meta, kw, g = load(name)
    cls = {'dict': cw.CraftingWorldEnv, 'int64': cw.CraftingWorldEnv, 'flat': cw.CraftingWorldEnvFlat,
           'onehot': cw.CraftingWorldEnvOneHot}.get(variant, cw.CraftingWorldEnvAltObs)
    exact = variant.startswith('alt_exact')
    extra = {}
    if variant == 'int64' or exact:
        extra['reference_dtypes'] = True
    if variant in ('alt_stacked', 'alt_exact_stacked'):
        extra['stacked_obs'] = True
    if variant == 'flat':
        kw = {k: v for k, v in kw.items() if k != 'fixed_init_state'}
    env = cls(**kw, **extra)
    env.set_rng_state(g['key0'], int(g['pos0']))

    def frames(o):      # -> (observation, desired_goal or None, init_observation or None) as uint8
        if variant == 'flat':
            return o, None, None
        if variant in ('alt_stacked', 'alt_exact_stacked'):
            assert o.shape[0] == 4 and np.array_equal(o[0], o[2])
            return o[0], o[1], o[3]
        return o['observation'], o['desired_goal'], o['init_observation']

    def grid_of(oh):
        return (oh[:, :, :8] * np.arange(1, 9)).sum(axis=2).astype(np.uint8)

    ri = 0

    def check_reset(o, t):
        nonlocal ri
        assert g['r_at_step'][ri] == t
        if variant == 'onehot':
            assert crc(grid_of(o['observation'])) == crc(g['r_grid'][ri]) and np.array_equal(o['observation'], o['init_observation'])
            assert tuple(np.argwhere(o['observation'][:, :, 8] == 1)[0]) == tuple(g['r_agent'][ri])
        else:
            ob, des, ini = frames(o)
            assert crc(ob.astype(np.uint8)) == g['r_obs_crc'][ri], (name, 'reset obs', ri)
            if des is not None:
                assert crc(des.astype(np.uint8)) == g['r_desired_img_crc'][ri] and crc(ini.astype(np.uint8)) == g['r_init_img_crc'][ri]
            if variant == 'int64' or exact:
                assert ob.dtype == np.int64
            if exact:
                assert crc(ob.astype(np.int16)) == g['r_obs_crc16'][ri] and crc(des.astype(np.int16)) == g['r_desired_img_crc16'][ri]
                assert crc(ini.astype(np.int16)) == g['r_init_img_crc16'][ri]
        bits = sum(int(b) << i for i, b in enumerate(env.desired_goal_vector[0]))
        assert bits == g['r_desired'][ri] and env.ep_no == g['r_ep_no'][ri]
        ri += 1

    check_reset(env.reset(), 0)
    T = len(g['action']) if exact else min(len(g['action']), 1500)
    over = 0
    for t in range(T):
        o, r, d, info = env.step(int(g['action'][t]))
        assert r == g['reward'][t] and d == bool(g['done'][t]), (name, t)
        if exact:
            ob = frames(o)[0]
            assert crc(ob.astype(np.int16)) == g['obs_crc16'][t] and int(ob.max()) == g['obs_max'][t], (name, 'exact obs', t)
            over += int(ob.max() > 255)
            if ob.max() > 255:
                assert np.array_equal(env.render(), ob)                  # render() of the current state is exact too
        assert sum(int(b) << i for i, b in enumerate(info['achieved_goal'][0])) == g['achieved'][t], (name, 'achieved', t)
        assert env.step_num == g['step_num'][t]
        if d:
            check_reset(env.reset(), t + 1)
        elif variant == 'onehot':
            assert crc(grid_of(o['observation'])) == g['grid_crc'][t], (name, 'grid', t)
            ar, ac = g['agent'][t]
            assert o['observation'][ar, ac, 8] == 1 and o['observation'][:, :, 8].sum() == 1
            assert o['observation'][ar, ac, 9:].sum() == (1 if g['hold'][t] else 0)
            if g['hold'][t]:
                assert o['observation'][ar, ac, 8 + g['hold'][t]] == 1
        else:
            assert crc(frames(o)[0].astype(np.uint8)) == g['obs_crc'][t], (name, 'obs', t)
    assert ri >= 2
    if exact:
        assert over == int((g['obs_max'] > 255).sum()) and (over == 9 if name == 'alt4_double' else True)
        if 'stacked' not in variant:
            assert np.array_equal(env.obs_image.astype(np.int16), g['final_obs16'])
    env.close()


@pytest.mark.gpu
@pytest.mark.parametrize('reference_dtypes', [True, False])
def test_altobs_stacked_obs_replays_the_reference_fixture(reference_dtypes):
    """CraftingWorldEnvAltObs(stacked_obs=True) (altobs.py:116-119, 258-261, 408-412): reset() and step() return ONE array, the four images
    stacked.  tests/golden/alt6_stacked.npz was captured from the reference class built with stacked_obs=True and holds the CRC of every array
    it returned (84 resets, 3 000 steps; int16 view): the facade must return the same stacks -- order, shape and values, exactly with
    reference_dtypes=True (int64), modulo 256 in the default uint8."""
    import gym_craftingworld_amd as cw
    meta, kw, g = load('alt6_stacked')
    assert meta['stacked_obs'] and meta['env'] == 'CraftingWorldEnvAltObs'
    env = cw.CraftingWorldEnvAltObs(**kw, stacked_obs=True, reference_dtypes=reference_dtypes)
    env.set_rng_state(g['key0'], int(g['pos0']))
    assert tuple(env.observation_space.shape) == tuple(g['stack_shape'])
    calls = iter(g['stack_crc16'])

    def check(o, wraps=False):
        assert isinstance(o, np.ndarray) and tuple(o.shape) == tuple(g['stack_shape']) and o.dtype == (np.int64 if reference_dtypes else np.uint8)
        want = next(calls)
        if reference_dtypes or not wraps:                       # (uint8 frames hold the reference's values modulo 256: compared where nothing exceeds 255)
            assert crc(o.astype(np.int16)) == want
    check(env.reset())
    for t in range(len(g['action'])):
        o, r, d, info = env.step(int(g['action'][t]))
        assert r == g['reward'][t] and d == bool(g['done'][t]), t
        check(o, wraps=int(g['obs_max'][t]) > 255)
        if d:
            check(env.reset())
    assert next(calls, None) is None                            # every returned array was compared
    env.close()


@pytest.mark.gpu
@pytest.mark.parametrize('N,size,raster', [(3000, 21, 'ray'), (65536, 21, 'ray'), (2000, 32, 'ray'), (700, 70, 'ray'), (3001, 10, 'ray'), (133, 128, 'ray'),
                                           (4099, 8, 'ray'), (5003, 5, 'ray'), (3333, 4, 'ray'), (2001, 9, 'ray'), (1999, 6, 'ray'), (777, 7, 'ray'),
                                           (3000, 21, 'alt'), (5000, 9, 'alt'), (65536, 21, 'alt'), (4001, 22, 'alt'), (1500, 32, 'alt'), (2999, 12, 'alt'),
                                           (3111, 4, 'alt'), (2777, 5, 'alt'), (1234, 8, 'alt'), (999, 11, 'alt'),
                                           (3001, 21, 'ray-chunked'), (70000, 21, 'ray-chunked'), (20000, 8, 'ray-chunked'), (30001, 5, 'ray-chunked'),
                                           (20011, 13, 'alt-chunked'), (30000, 5, 'alt-chunked'),
                                           # batches large enough for the four workgroups per CU that frames under 4 KiB are swept with, both painters
                                           (65536, 5, 'ray'), (50001, 7, 'ray'), (100003, 4, 'ray'), (40000, 8, 'ray'), (30011, 9, 'ray'), (60001, 6, 'alt'),
                                           (5003, 5, 'ray-nogather'), (2222, 7, 'ray-nogather')])
def test_full_frame_step_of_every_frame_size_equals_the_dirty_cell_engine(N, size, raster, monkeypatch):
    """The full-frame step = step kernel (finished envs take their look-ahead records), ONE sweep of aligned 4-KiB pieces over the
    observation array whatever the frame size (a piece overlaps two frames from 10x10 / AltObs 12x12 up, as many as nine of the smallest),
    then the done list's kernel (INIT_OBS / desired_goal frames, the resets of envs that found no record).  Every frame size, both
    rasters, batches that end in a partial piece, and large batches swept in chunks must leave exactly the frames, results and random
    streams of the dirty-cell engine, with episodes of 7 steps ending on every step (phases spread out) and all at once."""
    from gym_craftingworld_amd import CraftingWorldVecEnv
    if raster.endswith('-chunked'):  # large batches are swept in several launches over consecutive env ranges (cw_piece_chunks): here the chunks are 4 096 envs,
        raster = raster[:-8]         # the last one shorter
        monkeypatch.setenv('CW_TUNE_RENDER_CHUNK_ROUNDS', '1')
    gather = raster == 'ray' and size <= 7           # the smallest Ray frames: cw_render_gather_kernel (every lane computes its own 16-byte chunks)
    if raster.endswith('-nogather'):                  # ... and the piece sweep on the same frames
        raster, gather = raster[:-9], False
        monkeypatch.setenv('CW_TUNE_GATHER', '0')
    monkeypatch.setenv('CW_TUNE_PERIOD_NS', '700' if N % 2 else '0')       # (clocked and unclocked sweeps paint the same frames)
    kw = dict(size=(size, size), max_steps=7, seed=29, raster=raster)
    full = CraftingWorldVecEnv(N, obs_mode='pixels', **kw)
    assert full.render_kernel_name() == ('cw_render_gather_kernel' if gather else 'cw_render_pieces_kernel')
    dirty = CraftingWorldVecEnv(N, obs_mode='pixels_dirty', **kw)
    for e in (full, dirty):
        e.reset()
    for k in ('observation', 'desired_goal', 'init_observation'):
        assert torch.equal(full._observation()[k], dirty._observation()[k]), ('reset', k)
    gen = torch.Generator(device='cuda').manual_seed(5)
    for t in range(20):
        a = torch.randint(0, 6, (N,), device='cuda', dtype=torch.uint8, generator=gen)
        if t == 9:                                         # from here on the phases are spread out: env i has taken i % 7 steps
            for e in (full, dirty):
                e.set_state(step_num=(np.arange(N) % 7).astype(np.int32))
        od, rd, dd, _ = dirty.step(a)
        o, r, d, _ = full.step(a)
        assert torch.equal(o['observation'], od['observation']), (t, 'observation')
        assert torch.equal(o['desired_goal'], od['desired_goal']), (t, 'desired_goal')
        assert torch.equal(o['init_observation'], od['init_observation']), (t, 'init_observation')
        assert torch.equal(r, rd) and torch.equal(d, dd), t
        if t == 12:
            full._obs.fill_(9)                            # poison: a frame nobody paints would keep this value
    kd, pd = dirty.get_rng_states()
    k, p_ = full.get_rng_states()
    assert np.array_equal(k, kd) and np.array_equal(p_, pd)
    assert torch.equal(full.counters, dirty.counters)
    full.close()
    dirty.close()


@pytest.mark.gpu
@pytest.mark.parametrize('rate', ['9.0', '6.8', None])
def test_frames_do_not_depend_on_the_sweeps_clock(rate, monkeypatch):
    """The sweep's clock and its guard (cw_engine.cpp: sweep_guard_tick) only decide WHEN a piece of the observation array is written.  Started
    far above what the memory system takes (9 TB/s: the guard steps down), far below (6.8: it probes up) and at cw_create's own choice, 2 000 steps of the headline
    batch leave exactly the dirty-cell engine's frames, outputs, counters and RNG streams -- whatever the guard did on this box (what it did is
    tests/test_zz_perf_floors.py's business)."""
    from gym_craftingworld_amd import CraftingWorldVecEnv
    N, T, kw = 65536, 2000, dict(size=(21, 21), max_steps=300, seed=3)
    acts = torch.randint(0, 6, (64, N), device='cuda', dtype=torch.uint8, generator=torch.Generator(device='cuda').manual_seed(2))
    if rate:
        monkeypatch.setenv('CW_TUNE_RATE_TBS', rate)
    e = CraftingWorldVecEnv(N, obs_mode='pixels', **kw)
    monkeypatch.delenv('CW_TUNE_RATE_TBS', raising=False)
    dirty = CraftingWorldVecEnv(N, obs_mode='pixels_dirty', **kw)
    for env in (e, dirty):
        env.reset()
        env.set_state(step_num=((np.arange(N) * 7) % 300).astype(np.int32))      # (envs finish on every step)
    for t in range(T):
        e.step_async(acts[t % 64])
        dirty.step_async(acts[t % 64])
        if t % 500 == 499:
            for k in ('observation', 'desired_goal', 'init_observation'):
                assert torch.equal(e._observation()[k], dirty._observation()[k]), (t, k)
            assert torch.equal(e.reward, dirty.reward) and torch.equal(e.done, dirty.done)
    assert torch.equal(e.counters, dirty.counters)
    (ke, pe), (kd, pd_) = e.get_rng_states(), dirty.get_rng_states()
    assert np.array_equal(ke, kd) and np.array_equal(pe, pd_)
    e.close(); dirty.close()


@pytest.mark.parametrize('obs_mode,reward_style', [('state', None), ('pixels_dirty', 'subset')])
def test_episode_statistics_of_envs_stepped_past_done_without_auto_reset(obs_mode, reward_style):
    """An engine WITHOUT auto-reset keeps stepping a finished env until reset(), as the reference does (ray.py:367): a goal that stays satisfied pays
    MAX_STEPS again on every step that changes the state, time-outs stay done.  info['episode'] = {'r', 'l'} at every done step must be the SUM of
    the rewards the oracle returned since the reset, and the step count -- not the closed form of an episode that stopped at its first done."""
    from gym_craftingworld_amd import CraftingWorldVecEnv
    from oracle import OracleEnv
    N, T, kw = 384, 70, dict(size=(5, 5), max_steps=14, reward_style=reward_style, selected_tasks=['MoveAxe', 'MoveSticks', 'GoToHouse', 'EatBread'], number_of_tasks=1)
    keys, pos = _np_states(N, 31000)
    env = CraftingWorldVecEnv(N, obs_mode=obs_mode, auto_reset=False, **kw)
    env.set_rng_states(keys, pos)
    env.reset()
    oras = [OracleEnv(rng_state=(keys[i], int(pos[i])), **kw) for i in range(N)]
    for o in oras:
        o.reset()
    acts = np.random.RandomState(5).randint(0, 6, size=(T, N))
    ret = np.zeros(N, np.int64)
    repeated = checked = 0
    n_succ = np.zeros(N, np.int64)
    for t in range(T):
        _, rew, done, info = env.step(torch.as_tensor(acts[t], device=env.device))
        o_r = np.empty(N, np.int64)
        o_d = np.empty(N, bool)
        for i, o in enumerate(oras):
            _, o_r[i], o_d[i], _ = o.step(int(acts[t, i]))
        ret += o_r
        n_succ += o_r == kw['max_steps']
        assert np.array_equal(rew.cpu().numpy(), o_r) and np.array_equal(done.cpu().numpy(), o_d), t
        if o_d.any():
            assert np.array_equal(info['episode']['r'].cpu().numpy()[o_d], ret[o_d]), ('episode return', t)
            assert np.array_equal(info['episode']['l'].cpu().numpy()[o_d], np.full(N, t + 1)[o_d]), ('episode length', t)
            checked += int(o_d.sum())
            repeated += int((n_succ[o_d] >= 2).sum())
    assert checked > N * (T - 14) and repeated > 50          # (every env is done from step 14 on; many were paid more than once)
    env.close()


@pytest.mark.gpu
def test_full_frame_soak_equals_dirty_cell_engine(monkeypatch):
    """3 000 steps of 65 536 full-frame envs with the episode phases spread out (~220 envs finish on every step and take their look-ahead
    records; the refill kernel runs every max_steps/4 steps, 8 ... 64): every 250 steps all three frames, and at the end results, counters and random
    streams, must equal the dirty-cell engine's."""
    from gym_craftingworld_amd import CraftingWorldVecEnv
    N, T = 65536, 3000
    kw = dict(size=(21, 21), max_steps=300, seed=77)
    full = CraftingWorldVecEnv(N, obs_mode='pixels', **kw)
    dirty = CraftingWorldVecEnv(N, obs_mode='pixels_dirty', **kw)
    for e in (full, dirty):
        e.reset()
        e.set_state(step_num=((np.arange(N) * 7) % 300).astype(np.int32))
    gen = torch.Generator(device='cuda').manual_seed(11)
    acts = torch.randint(0, 6, (64, N), device='cuda', dtype=torch.uint8, generator=gen)
    for t in range(T):
        of, rf, df, _ = full.step(acts[t % 64])
        od, rd, dd, _ = dirty.step(acts[t % 64])
        if t % 250 == 249 or t == T - 1:
            assert torch.equal(rf, rd) and torch.equal(df, dd), t
            for k in ('observation', 'desired_goal', 'init_observation'):
                assert torch.equal(of[k], od[k]), (t, k)
    assert torch.equal(full.counters, dirty.counters)
    kf, pf = full.get_rng_states()
    kd, pd_ = dirty.get_rng_states()
    assert np.array_equal(kf, kd) and np.array_equal(pf, pd_)
    full.close(); dirty.close()


@pytest.mark.gpu
def test_long_reset_chains_on_one_stream_vs_oracle():
    """3 000 consecutive episodes on each of 256 MT19937 streams (max_steps=1: every step ends an episode), so that
    the lazily regenerated state is entered at every alignment and wraps hundreds of times per stream; state, goal
    state and the exact stream (key and position) must still equal the oracle's at the end and on the way."""
    from gym_craftingworld_amd import CraftingWorldVecEnv
    from oracle import OracleBatch
    N, T, chunk = 256, 3000, 750
    kw = dict(size=(9, 9), max_steps=1)
    env = CraftingWorldVecEnv(N, obs_mode='state', seed=2024, **kw)
    keys, pos = env.get_rng_states()
    ora = OracleBatch(N, rng_states=[(keys[i], int(pos[i])) for i in range(N)], **kw)
    env.reset()
    ora.reset()
    rng = np.random.RandomState(8)
    for c in range(T // chunk):
        acts = rng.randint(0, 6, size=(chunk, N)).astype(np.int8)
        if c % 2:                                                # alternate the two code paths that reset inline
            env.rollout(torch.as_tensor(acts.astype(np.uint8), device=env.device), record=False)
        else:
            for t in range(chunk):
                env.step(torch.as_tensor(acts[t].astype(np.uint8), device=env.device))
        assert ora.rollout(acts, nthreads=8) == chunk * N
        st = env.get_state()
        k2, p2 = env.get_rng_states()
        for i, s in enumerate(ora.states()):
            assert np.array_equal(st['grid'][i], s['grid']) and np.array_equal(st['goal_grid'][i], s['goal_grid']), (c, i)
            assert st['desired'][i] == s['desired'] and st['ep_no'][i] == s['ep_no'], (c, i)
        for i in range(0, N, 5):
            ok, op = ora.envs[i].get_rng()
            assert p2[i] % 624 == op % 624, (c, i)
            if op % 624:
                assert np.array_equal(k2[i][1:], ok[1:]), (c, i)
    assert int(env.counters[1].item()) == N * T
    env.close()


@pytest.mark.gpu
def test_random_configurations_vs_oracle():
    """A seeded sweep over the configuration space itself -- grid size, episode length, ordered task subsets with
    number_of_tasks / stacking / reward style, fixed-init pools, observation mode, raster -- each drawn at random and run
    against the oracle with auto-reset: reward, done, masks every step; state, frames and RNG stream at the end."""
    from gym_craftingworld_amd import CraftingWorldVecEnv
    from oracle import OracleBatch
    cfg_rng = np.random.RandomState(20260101)
    for c in range(40):
        S = int(cfg_rng.choice([4, 5, 6, 7, 9, 11, 13, 16, 21, 24, 33]))
        sel = [TASKS[i] for i in cfg_rng.permutation(9)[:cfg_rng.randint(1, 10)]]
        kw = dict(size=(S, S), max_steps=int(cfg_rng.randint(1, 40)), selected_tasks=sel,
                  number_of_tasks=int(cfg_rng.randint(1, len(sel) + 1)) if cfg_rng.rand() < 0.7 else None,
                  stacking=bool(cfg_rng.rand() < 0.7), reward_style=None if cfg_rng.rand() < 0.6 else 'subset',
                  fixed_init_state=int(cfg_rng.choice([0, 0, 0, 1, 5])))
        obs_mode = str(cfg_rng.choice(['pixels', 'pixels_dirty', 'state']))
        raster = 'alt' if (obs_mode != 'state' and cfg_rng.rand() < 0.3) else 'ray'
        N, T = int(cfg_rng.choice([1, 3, 17, 64, 130])), int(cfg_rng.randint(20, 90))
        tag = (c, S, kw, obs_mode, raster, N, T)
        keys, pos = _np_states(N, 5000 + 37 * c)
        env = CraftingWorldVecEnv(N, obs_mode=obs_mode, raster=raster, **kw)
        env.set_rng_states(keys, pos)
        if kw['fixed_init_state']:
            from gym_craftingworld_amd import _lib as L
            L.check(env._lib.cw_generate_fixed_states(env._h, env._stream()), 'pool')
        ora = OracleBatch(N, rng_states=list(zip(keys, pos)), alt_obs=(raster == 'alt'), **kw)
        env.reset(); ora.reset()
        acts = np.random.RandomState(c).randint(0, 6, size=(T, N)).astype(np.int64)
        for t in range(T):
            obs, rew, done, info = env.step(torch.as_tensor(acts[t], device=env.device))
            o_rew, o_done, o_ach = np.empty(N, np.int32), np.zeros(N, bool), np.empty(N, np.int64)
            for i, e in enumerate(ora.envs):
                _, o_rew[i], o_done[i], _ = e.step(int(acts[t, i]))
                o_ach[i] = e.view().achieved
                if o_done[i]:
                    e.reset()
            assert np.array_equal(rew.cpu().numpy(), o_rew) and np.array_equal(done.cpu().numpy(), o_done), (tag, t)
            assert np.array_equal(info['achieved_goal'].cpu().numpy().astype(np.int64) & 0xFFFF, o_ach), (tag, t)
        st = env.get_state()
        frames = env.render().cpu().numpy()
        k2, p2 = env.get_rng_states()
        for i, s in enumerate(ora.states()):
            assert np.array_equal(st['grid'][i], s['grid']) and np.array_equal(st['goal_grid'][i], s['goal_grid']), (tag, i)
            assert tuple(st['agent_rc'][i]) == s['agent'] and st['hold'][i] == s['hold'] and st['desired'][i] == s['desired'], (tag, i)
            assert st['ep_no'][i] == s['ep_no'] and st['step_num'][i] == s['step_num'], (tag, i)
            assert np.array_equal(frames[i], s['obs']), (tag, i)
            if obs_mode != 'state':
                assert np.array_equal(obs['observation'][i].cpu().numpy(), s['obs']), (tag, i)
                assert np.array_equal(obs['desired_goal'][i].cpu().numpy(), s['desired_img']), (tag, i)
            ok, op = ora.envs[i].get_rng()
            assert p2[i] % 624 == op % 624, (tag, i)
        env.close()


@pytest.mark.gpu
def test_facade_render_of_a_supplied_state():
    """render(state=one_hot) (ray.py:442-520 with a caller-supplied state): equals the frame of an env that is in that
    state, for states visited along a trajectory (held items included), and leaves the env itself untouched."""
    import gym_craftingworld_amd as cw
    env = cw.make('craftingworld-v3', size=(7, 7), max_steps=60)
    env.seed(5)
    other = cw.make('craftingworld-v3', size=(7, 7), max_steps=60)
    other.seed(99)
    env.reset(); other.reset()
    rng = np.random.RandomState(4)
    for t in range(60):
        obs, _, d, _ = env.step(int(rng.choice([0, 1, 2, 3, 4, 4, 5])))
        oh = env.obs_one_hot
        before = other.obs_image.copy()
        assert np.array_equal(other.render(state=oh), obs['observation']), t
        assert np.array_equal(other.obs_image, before) and np.array_equal(other.render(), before)
        if d:
            break
    st = env.get_state()
    checked = 0
    for hold in (1, 2, 3):                                   # a held item shows on the agent's tile (ray.py:484-486)
        g2 = st['grid'][0].copy()
        where = np.argwhere(g2 == hold)                      # codes 1,2,3 = sticks, axe, hammer = hold ids
        if len(where) == 0:
            continue
        g2[tuple(where[0])] = 0                              # ... picked up: no longer on the grid
        env.set_state(grid=g2[None], init_grid=st['init_grid'], agent_rc=st['agent_rc'], hold=np.array([hold]))
        oh = env.obs_one_hot
        assert oh[:, :, 8 + hold].sum() == 1
        assert np.array_equal(other.render(state=oh), env.render()), hold
        checked += 1
    assert checked >= 2
    with pytest.raises(ValueError):
        other.render(state=np.zeros((5, 5, 12), int))
    env.close(); other.close()


@pytest.mark.gpu
def _reference_alt_render_of_any_state(state):
    """craftingworld_altobs.py:489-560 restated in numpy for a caller-supplied one-hot state (test-side oracle): pixel k of a cell's
    3x3 tile = CPV_COLORS[k] x (channel k, + hold channel 9 + k for k < 3); three more rows, pixels 3..5 white if anything is held."""
    cpv = np.array([(45, 82, 160), (255, 102, 102), (204, 204, 0), (211, 211, 211), (34, 133, 34), (0, 215, 255), (153, 52, 255),
                    (10, 215, 100), (0, 0, 255)], dtype=np.int64)                                   # CPV_COLORS, altobs.py:26-27
    h, w = state.shape[:2]
    items = state[:, :, :9].astype(np.int64).copy()
    items[:, :, :3] += state[:, :, 9:12]
    img = np.zeros((3 * h + 3, 3 * w, 3), dtype=np.int64)
    for k in range(9):
        img[k // 3:3 * h:3, k % 3::3] = items[:, :, k][:, :, None] * cpv[k]
    if state[:, :, 9:12].max() > 0:
        img[3 * h:, 3:6] = 255
    return img


def test_altobs_facade_render_of_a_supplied_state():
    """CraftingWorldEnvAltObs.render(state=one_hot) returns the AltObs image ((S+1)*3, S*3, 3) of THAT state (round 2 returned a Ray-style
    image here): equal to the frame of an env that is in that state along a trajectory with held items, equal to a numpy restatement of
    craftingworld_altobs.py:489-560 for arbitrary states (objects doubled up, hold flags anywhere), exact with reference_dtypes=True."""
    import gym_craftingworld_amd as cw
    env = cw.CraftingWorldEnvAltObs(size=(6, 6), max_steps=80, reference_dtypes=True)
    other = cw.CraftingWorldEnvAltObs(size=(6, 6), max_steps=80, reference_dtypes=True)
    env.seed(3); other.seed(8)
    env.reset(); other.reset()
    rng = np.random.RandomState(12)
    held = 0
    for t in range(80):
        obs, _, d, _ = env.step(int(rng.choice([0, 1, 2, 3, 4, 4, 5])))
        oh = env.obs_one_hot
        held += int(oh[:, :, 9:].sum() > 0)
        got = other.render(state=oh)
        assert got.shape == (21, 18, 3) and got.dtype == np.int64
        assert np.array_equal(got, obs['observation']) and np.array_equal(got, _reference_alt_render_of_any_state(oh)), t
        assert np.array_equal(env.render(), obs['observation'])
        if d:
            break
    st = env.get_state()
    held = 0
    for hold in (1, 2, 3):                                   # held items: counted on their own object pixel at the agent's tile + the strip's flag
        g2 = st['grid'][0].copy()
        where = np.argwhere(g2 == hold)
        if len(where) == 0:
            continue
        g2[tuple(where[0])] = 0
        if hold == 1:                                        # ... and sticks held OVER sticks: 2 x (45, 82, 160) = (90, 164, 320)
            g2[tuple(st['agent_rc'][0])] = 1
        env.set_state(grid=g2[None], init_grid=st['init_grid'], agent_rc=st['agent_rc'], hold=np.array([hold]))
        oh = env.obs_one_hot
        want = _reference_alt_render_of_any_state(oh)
        assert oh[:, :, 8 + hold].sum() == 1 and (want.max() == 320) == (hold == 1)
        assert np.array_equal(other.render(state=oh), want) and np.array_equal(env.render(), want), hold
        held += 1
    assert held >= 2
    for trial in range(20):                                  # arbitrary states: the reference renders whatever the array holds
        st = (rng.rand(6, 6, 12) < 0.15).astype(int)
        st[:, :, 8] = 0
        st[rng.randint(6), rng.randint(6), 8] = 1
        if trial % 3 == 0:
            st[:, :, 9:] = 0
        assert np.array_equal(other.render(state=st), _reference_alt_render_of_any_state(st)), trial
    with pytest.raises(IndexError):
        other.render(state=np.zeros((6, 6, 12), int))       # no agent: state_idxs[0][0], altobs.py:501
    env.close(); other.close()


@pytest.mark.parametrize('cls_name,kw', [('CraftingWorldEnv', dict(size=(7, 7), max_steps=40)), ('CraftingWorldEnv', dict()),
                                         ('CraftingWorldEnvFlat', dict()), ('CraftingWorldEnvAltObs', dict(size=(6, 6), max_steps=30)),
                                         ('CraftingWorldEnv', dict(size=(5, 5), max_steps=25, reference_dtypes=True)),
                                         ('CraftingWorldEnvOneHot', dict(size=(6, 6), max_steps=30)), ('CraftingWorldEnvOneHot', dict())])
def test_resident_stepper_equals_the_launch_path(cls_name, kw):
    """step() of the N=1 classes rings a resident kernel's doorbell instead of launching a kernel (cw_step_resident): same observations,
    rewards, dones, masks, state, counters and random streams as the launch + stream-sync path (resident=False), step for step over many
    episodes -- with the kernel idling out between steps (a 5-ms pause: it leaves after 2 ms and is started again), parked by other calls
    in mid-episode (render(), obs_one_hot, the RNG state) and by every reset()."""
    import time
    import gym_craftingworld_amd as cw
    cls = getattr(cw, cls_name)
    a_env, b_env = cls(resident=True, **kw), cls(resident=False, **kw)
    assert a_env._resident and not b_env._resident
    for e in (a_env, b_env):
        e.seed(123)
    rng = np.random.RandomState(2)

    def same(x, y):
        if isinstance(x, dict):
            return all(np.array_equal(x[k], y[k]) for k in x)
        return np.array_equal(x, y)

    assert same(a_env.reset(), b_env.reset())
    episodes = 0
    for t in range(900):
        a = int(rng.randint(6))
        oa, ra, da, ia = a_env.step(a)
        ob, rb, db, ib = b_env.step(a)
        assert ra == rb and da == db and same(oa, ob), t
        assert np.array_equal(ia['achieved_goal'], ib['achieved_goal']) and np.array_equal(ia['desired_goal'], ib['desired_goal']), t
        if t % 97 == 5:
            time.sleep(0.005)                                    # the resident kernel idles out (2 ms) and is started again by the next step
        if t % 131 == 7:                                         # other entry points park it in mid-episode
            assert np.array_equal(a_env.render(), b_env.render()) and np.array_equal(a_env.obs_one_hot, b_env.obs_one_hot)
            assert a_env.agent_pos == b_env.agent_pos
        if da:
            episodes += 1
            ka, pa = a_env.get_rng_state()
            kb, pb = b_env.get_rng_state()
            assert pa == pb and np.array_equal(ka[1:], kb[1:])
            assert same(a_env.reset(), b_env.reset())
    assert episodes >= 3
    assert torch.equal(a_env._vec.counters, b_env._vec.counters) and torch.equal(a_env._vec.hdr, b_env._vec.hdr)
    with pytest.raises(IndexError):
        a_env.step(6)
    a_env.close(); b_env.close()


def test_facade_store_gif_files_and_rng_draw(tmp_path, monkeypatch):
    """store_gif=True on the single-env class (ray.py:142-143,160-167,205-216,370-374,769-782): the env id is one
    randint(0, 1000000) taken from the env's own stream at construction (so later placements shift exactly as the
    reference's do), and every render_save_rate-th finished episode is written as
    renders/env<id>/E<ep>(<steps>)_<desired>(<achieved>).gif with one frame per reset/step."""
    import glob
    from PIL import Image
    import gym_craftingworld_amd as cw
    monkeypatch.chdir(tmp_path)
    kw = dict(size=(5, 5), max_steps=6, seed=3)
    plain = cw.CraftingWorldEnv(**kw)
    rec = cw.CraftingWorldEnv(store_gif=True, render_save_rate=2, **kw)
    k, p = plain.get_rng_state()
    rs = np.random.RandomState()
    rs.set_state(('MT19937', k, p, 0, 0.0))
    assert rec.env_id == int(rs.randint(0, 1000000))
    k2, p2 = rec.get_rng_state()
    assert p2 == rs.get_state()[2] and np.array_equal(k2[1:], rs.get_state()[1][1:])   # (word 0's low bits do not survive export)
    assert os.path.isdir('renders/env%d' % rec.env_id)
    rng = np.random.RandomState(0)
    rec.reset()
    episodes = 0
    while episodes < 5:
        _, _, d, _ = rec.step(int(rng.randint(6)))
        if d:
            rec.reset()
            episodes += 1
    files = sorted(glob.glob('renders/env%d/*.gif' % rec.env_id))
    eps = sorted(int(os.path.basename(f)[1:].split('(')[0]) for f in files)
    assert eps == [0, 2, 4], files
    im = Image.open([f for f in files if os.path.basename(f).startswith('E0(')][0])
    assert im.size == ((2 * 20 + 4) * 4, 20 * 4) and 1 <= im.n_frames <= 7
    plain.close(); rec.close()
    # CraftingWorldEnvFlat writes a finished episode's GIF only if it achieved something or every 30th episode (craftingworld_flat.py:64-71)
    flat = cw.CraftingWorldEnvFlat(size=(5, 5), max_steps=4, seed=4, store_gif=True, render_save_rate=1)
    flat.reset()
    episodes, achieved = 0, {}
    while episodes < 33:
        _, _, d, info = flat.step(int(rng.randint(6)))
        if d:
            achieved[flat.ep_no] = bool(info['achieved_goal'].any())
            flat.reset()
            episodes += 1
    eps = sorted(int(os.path.basename(f)[1:].split('(')[0]) for f in glob.glob('renders/env%d/*.gif' % flat.env_id))
    assert eps == sorted(e for e, a in achieved.items() if a or e % 30 == 0), (eps, achieved)
    assert 0 in eps and 30 in eps and len(eps) < 33
    flat.close()


@pytest.mark.gpu
def test_one_hot_of_goal_and_init_states_vs_oracle():
    """The OneHot variant's other two observations at batch scale: one_hot(which='goal') is imagine_obs' final state
    (onehot.py:310), one_hot(which='init') the state at reset (onehot.py:203), both [N,S,S,12] on the device."""
    from gym_craftingworld_amd import CraftingWorldVecEnv
    from oracle import OracleBatch
    N, kw = 96, dict(size=(8, 8), max_steps=15)
    keys, pos = _np_states(N, 777)
    env = CraftingWorldVecEnv(N, obs_mode='state', **kw)
    env.set_rng_states(keys, pos)
    ora = OracleBatch(N, rng_states=list(zip(keys, pos)), **kw)
    env.reset(); ora.reset()
    acts = np.random.RandomState(1).randint(0, 6, size=(40, N)).astype(np.int8)
    env.rollout(torch.as_tensor(acts.astype(np.uint8), device=env.device), record=False)
    ora.rollout(acts, nthreads=4)
    goal, init, cur = (env.one_hot(which=w).cpu().numpy() for w in ('goal', 'init', 'current'))
    codes = lambda oh: (oh[:, :, :8] * np.arange(1, 9)).sum(axis=2)   # noqa: E731
    for i, s in enumerate(ora.states()):
        assert np.array_equal(codes(goal[i]), s['goal_grid']) and np.array_equal(codes(init[i]), s['init_grid']), i
        assert np.array_equal(codes(cur[i]), s['grid']), i
        assert tuple(np.argwhere(goal[i][:, :, 8] == 1)[0]) == s['goal_agent'], i
        assert goal[i][:, :, 8].sum() == 1 and init[i][:, :, 8].sum() == 1 and goal[i][:, :, 9:].sum() == 0 and init[i][:, :, 9:].sum() == 0
    env.close()


@pytest.mark.gpu
@pytest.mark.parametrize('obs_mode,raster,pool', [('pixels', 'ray', 0), ('pixels_dirty', 'alt', 0), ('state', 'ray', 0), ('state', 'ray', 3)])
def test_checkpoint_resume_is_bit_identical(obs_mode, raster, pool, tmp_path):
    """save_checkpoint() in the middle of episodes, load_checkpoint() into a fresh engine (other seed -- so other
    fixed_init_state pools --, other history, or no reset at all): from there on rewards, dones, masks, all three frames,
    the raw state tensors (hdr, slot_pos: the state-mode observation), counters and RNG streams equal the run that never
    stopped.  Mixed task menus travel with the records; an engine with other menus / shape refuses the file."""
    from gym_craftingworld_amd import CraftingWorldVecEnv
    N = 300
    menus = [dict(), dict(selected_tasks=['ChopTree', 'MoveAxe', 'EatBread'], number_of_tasks=2, reward_style='subset')]
    env_menu = (np.arange(N) % 2).astype(np.uint8)
    kw = dict(size=(9, 9), max_steps=23, obs_mode=obs_mode, raster=raster, fixed_init_state=pool, task_menus=menus)
    a = CraftingWorldVecEnv(N, seed=1, env_menu=env_menu, **kw)
    a.reset()
    gen = torch.Generator(device='cuda').manual_seed(6)
    acts = torch.randint(0, 6, (100, N), device='cuda', dtype=torch.uint8, generator=gen)
    for t in range(37):
        a.step(acts[t])
    path = str(tmp_path / 'ckpt')                     # written exactly there: no extension is appended
    a.save_checkpoint(path)
    assert os.path.isfile(path)
    b = CraftingWorldVecEnv(N, seed=999, env_menu=env_menu[::-1].copy(), **kw)      # (its own menu assignment is overwritten)
    if pool == 0:
        b.reset()
        for t in range(5):
            b.step(acts[90 + t])
    b.load_checkpoint(path)                           # pool > 0: straight into an engine that was never reset
    assert torch.equal(a.hdr, b.hdr) and torch.equal(a.slot_pos, b.slot_pos) and torch.equal(a.counters, b.counters)
    # the engine's private word -- the finished count the last sweep of the observation array saw -- travels too: the resumed engine's first sweep
    # sees the step it follows as the uninterrupted run's would (round 4 restored 4 of the 5 words: `finished` = counters[1] - 0, a storm launch)
    assert int(b._counters_raw[4]) <= int(b.counters[1])
    if obs_mode != 'state':                           # (the load's own repaint of the observation array has brought it up to date)
        assert int(b._counters_raw[4]) == int(b.counters[1])
    else:
        assert int(b._counters_raw[4]) == int(a._counters_raw[4])
    assert torch.equal(a.episode_return, b.episode_return) and torch.equal(a.episode_length, b.episode_length)
    if obs_mode != 'state':
        oa, ob = a._observation(), b._observation()
        for k in oa:
            assert torch.equal(oa[k], ob[k]), ('after load', k)
    for t in range(37, 90):
        oa, ra, da, ia = a.step(acts[t])
        ob, rb, db, ib = b.step(acts[t])
        assert torch.equal(ra, rb) and torch.equal(da, db), t
        assert torch.equal(ia['achieved_goal'], ib['achieved_goal']) and torch.equal(ia['desired_goal'], ib['desired_goal']), t
        for k in oa:                                  # state mode: the observation IS hdr / slot_pos
            assert torch.equal(oa[k], ob[k]), (t, k)
        assert torch.equal(a.hdr, b.hdr) and torch.equal(a.slot_pos, b.slot_pos), t
    assert int(a.counters[1]) > N                     # episodes ended (and, with a pool, were re-drawn from it) after the resume
    sa, sb = a.get_state(), b.get_state()
    for k in sa:
        assert np.array_equal(sa[k], sb[k]), k
    ka, pa = a.get_rng_states(); kb, pb = b.get_rng_states()
    assert np.array_equal(pa, pb) and np.array_equal(ka[:, 1:], kb[:, 1:])
    assert torch.equal(a.counters, b.counters)
    with pytest.raises(ValueError):
        CraftingWorldVecEnv(N + 1, seed=0, **kw).load_checkpoint(path)
    other_menus = [dict(), dict(selected_tasks=['ChopTree', 'MoveAxe', 'EatBread'], number_of_tasks=3, reward_style='subset')]
    with pytest.raises(ValueError):
        CraftingWorldVecEnv(N, seed=0, env_menu=env_menu, **dict(kw, task_menus=other_menus)).load_checkpoint(path)
    with open(path, 'rb') as f:
        blob = f.read()
    (tmp_path / 'short').write_bytes(blob[:len(blob) // 2])
    with pytest.raises(ValueError):
        b.load_checkpoint(str(tmp_path / 'short'))
    a.close(); b.close()


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['alt4_double', 'ray5_scripted'])
def test_batch_render_exact_is_the_references_int_image(name):
    """CraftingWorldVecEnv.render_exact(): the reference's int image of the current state through the BATCH class -- for the AltObs raster including the
    one pixel the uint8 frames cannot hold (alt4_double: sticks held over sticks on 9 steps, values up to 320; the fixture's int16 CRCs are the
    reference's own), for the Ray raster simply the uint8 frame."""
    from gym_craftingworld_amd import CraftingWorldVecEnv
    meta, kw, g = load(name)
    alt = meta['env'] == 'CraftingWorldEnvAltObs'
    env = CraftingWorldVecEnv(1, obs_mode='pixels_dirty', auto_reset=False, raster='alt' if alt else 'ray', **kw)
    env.set_rng_states(g['key0'][None], np.array([int(g['pos0'])]))
    env.reset()
    over = 0
    T = len(g['action']) if alt else 1200
    for t in range(T):
        _, r, d, _ = env.step(torch.as_tensor(g['action'][t:t + 1].astype(np.int32), device=env.device))
        assert int(r[0]) == g['reward'][t]
        img = env.render_exact()[0].cpu().numpy()
        assert img.dtype == np.int16
        if alt:
            assert crc(img) == g['obs_crc16'][t] and int(img.max()) == g['obs_max'][t], (name, t)
            over += int(img.max() > 255)
        else:
            assert crc(img.astype(np.uint8)) == g['obs_crc'][t] and img.max() <= 255, (name, t)
        assert np.array_equal(img.astype(np.uint8), env._obs[0].cpu().numpy())        # (the engine's frame is this image modulo 256)
        if bool(d[0]):
            env.reset()
    assert over == (int((g['obs_max'] > 255).sum()) if alt else 0) and over == (9 if alt else 0)
    env.close()


@pytest.mark.gpu
def test_facade_render_of_states_with_duplicated_objects():
    """render(state=...) of reachable states that hold an object TWICE (after ChopTree: two sticks; MakeBread: two breads;
    BuildHouse: two houses) -- every imagine_obs goal state for those tasks, e.g. the OneHot env's desired_goal.  The
    reference's render(state) (ray.py:442-486) accepts any one-hot; the frame must equal the engine's own goal frame."""
    import gym_craftingworld_amd as cw
    from gym_craftingworld_amd import CraftingWorldVecEnv
    S = 7
    dup_seen = set()
    face = cw.make('craftingworld-v3', size=(S, S))
    for name, code in (('ChopTree', 1), ('MakeBread', 6), ('BuildHouse', 7)):
        vec = CraftingWorldVecEnv(32, size=(S, S), obs_mode='pixels', seed=11, selected_tasks=[name], number_of_tasks=1)
        obs = vec.reset()
        goal_oh = vec.one_hot(which='goal').cpu().numpy()
        goal_img = obs['desired_goal'].cpu().numpy()
        for i in range(8):
            if goal_oh[i][:, :, code - 1].sum() == 2:
                dup_seen.add(name)
            assert np.array_equal(face.render(state=goal_oh[i].astype(int)), goal_img[i]), (name, i)
        vec.close()
    assert dup_seen == {'ChopTree', 'MakeBread', 'BuildHouse'}
    face.close()


@pytest.mark.parametrize('keep_terminal', [False, True])
def test_headline_shape_reset_storm_65536(keep_terminal):
    """The headline shape (65 536 envs, 21x21, full-frame pixel obs) across steps on which EVERY env times out at once
    (max_steps=20, T=45: storms at t=19 and t=39 -- ray.py:367 then reset ray.py:156-218 for the whole batch: on the first every env takes its
    look-ahead record, on the second -- four steps after a refill -- too): full frames == dirty-cell frames on the storm step itself and after
    it, a sample of 256 envs (the first 128, a stride through the batch, the last 64) == the CPU oracle (all three frames, terminal frame, RNG
    stream position), counters."""
    from gym_craftingworld_amd import CraftingWorldVecEnv
    from oracle import OracleBatch
    N, T, M = 65536, 45, 256
    kw = dict(size=(21, 21), max_steps=20)
    full = CraftingWorldVecEnv(N, obs_mode='pixels', seed=321, keep_terminal_obs=keep_terminal, **kw)
    dirty = CraftingWorldVecEnv(N, obs_mode='pixels_dirty', seed=321, keep_terminal_obs=keep_terminal, **kw)
    keys, pos = full.get_rng_states()
    idx = list(range(128)) + list(range(300, N - 64, (N - 364) // 64))[:64] + list(range(N - 64, N))
    assert len(idx) == M and idx[-1] == N - 1
    idx_t = torch.as_tensor(idx, device='cuda')
    ora = OracleBatch(M, rng_states=[(keys[i], int(pos[i])) for i in idx], **kw)
    full.reset(); dirty.reset(); ora.reset()
    gen = torch.Generator(device='cuda').manual_seed(77)
    dones = storms = 0
    for t in range(T):
        a = torch.randint(0, 6, (N,), device='cuda', dtype=torch.uint8, generator=gen)
        of, rf, df, inf = full.step(a)
        od, rd, dd, ind = dirty.step(a)
        assert torch.equal(rf, rd) and torch.equal(df, dd), t
        n_done = int(df.sum().item())
        dones += n_done
        an = a[idx_t].cpu().numpy()
        term_ref = {}
        o_done = np.zeros(M, bool)
        o_rew = np.zeros(M, np.int32)
        for i, e in enumerate(ora.envs):
            o, o_rew[i], o_done[i], _ = e.step(int(an[i]))
            if o_done[i]:
                term_ref[i] = o['observation'].copy()
                e.reset()
        assert np.array_equal(rf[idx_t].cpu().numpy(), o_rew) and np.array_equal(df[idx_t].cpu().numpy(), o_done), t
        storm = n_done > N * 0.9
        if storm or t % 13 == 0 or t == T - 1:
            storms += int(storm)
            for k in ('observation', 'desired_goal', 'init_observation'):
                assert torch.equal(of[k], od[k]), (t, k)
            assert torch.equal(full.render(), of['observation']), t
            if keep_terminal:
                d = df.nonzero().flatten()
                assert torch.equal(inf['terminal_observation'][d], ind['terminal_observation'][d]), t
                term = inf['terminal_observation'][idx_t].cpu().numpy()
                for i, fr in term_ref.items():
                    assert np.array_equal(term[i], fr), (t, i)
            fo, fg, fi = (of[k][idx_t].cpu().numpy() for k in ('observation', 'desired_goal', 'init_observation'))
            for i, s in enumerate(ora.states()):
                assert np.array_equal(fo[i], s['obs']) and np.array_equal(fg[i], s['desired_img']) and np.array_equal(fi[i], s['init_img']), (t, idx[i])
    assert storms == 2 and dones >= 2 * N
    _, p2 = full.get_rng_states()
    for i, e in enumerate(ora.envs):
        assert p2[idx[i]] % 624 == e.get_rng()[1] % 624, idx[i]
    assert torch.equal(full.hdr, dirty.hdr) and torch.equal(full.slot_pos, dirty.slot_pos)
    assert int(full.counters[1].item()) == dones == int(dirty.counters[1].item()) and int(full.counters[0].item()) == N * T
    full.close(); dirty.close()


def test_vector_env_surface_on_the_device():
    """gym.vector surface: batched observation_space (leading N on every Box) beside the single one,
    seed(int) -> env i seeded seed+i, seed(list) -> per-env seeds, reset_async/reset_wait, step_async/step_wait, closed."""
    from gym_craftingworld_amd import CraftingWorldVecEnv
    N = 48
    env = CraftingWorldVecEnv(N, size=(6, 6), max_steps=9, obs_mode='pixels', seed=5)
    assert env.is_vector_env and not env.closed and env.num_envs == N
    assert env.single_observation_space['observation'].shape == (24, 24, 3)
    assert env.observation_space['observation'].shape == (N, 24, 24, 3) and env.observation_space['desired_goal'].dtype == np.uint8
    assert env.action_space.nvec.tolist() == [6] * N and env.single_action_space.n == 6
    assert env.seed(100) == list(range(100, 100 + N))
    env.reset_async()
    a = {k: v.clone() for k, v in env.reset_wait().items()}
    seeds = [100 + i for i in range(N)]
    assert env.seed(seeds) == seeds                       # the same seeds as a list: the same episodes
    b = env.reset()
    for k in a:
        assert a[k].shape == env.observation_space[k].shape and torch.equal(a[k], b[k]), k
    perm = seeds[::-1]
    env.seed(np.array(perm))                              # env i now has env (N-1-i)'s stream
    c = env.reset()
    assert torch.equal(c['observation'], a['observation'].flip(0)) and torch.equal(c['desired_goal'], a['desired_goal'].flip(0))
    ks, ps = env.get_rng_states()
    ref = np.random.RandomState(perm[3])
    st = np.random.RandomState(); st.set_state(('MT19937', ks[3], int(ps[3]), 0, 0.0))
    burn = ref.randint(0, 2**32, size=2000, dtype=np.uint32)       # env 3's stream is RandomState(perm[3])'s, further along
    nxt = st.randint(0, 2**32, size=8, dtype=np.uint32)
    assert any(np.array_equal(burn[j:j + 8], nxt) for j in range(len(burn) - 8))
    with pytest.raises(ValueError):
        env.seed([1, 2, 3])
    with pytest.raises(RuntimeError):
        env.reset_wait()
    env.step_async(torch.zeros(N, dtype=torch.uint8, device=env.device))
    obs, rew, done, info = env.step_wait()
    assert rew.shape == (N,) and done.dtype == torch.bool
    st_env = CraftingWorldVecEnv(N, size=(6, 6), obs_mode='state', seed=5)
    assert st_env.observation_space['hdr'].shape == (N, 16) and st_env.observation_space['slot_pos'].shape == (N, 8)
    st_env.close()
    env.close()
    assert env.closed
    env.close()                                           # idempotent


@pytest.mark.parametrize('name', [n for n in fixture_names() if n.startswith(('flat', 'onehot'))])
def test_flat_and_onehot_facades_replay_their_own_reference_fixtures(name):
    """SURVEY 8f rank 1, pinned by the reference's own classes: fixtures captured from CraftingWorldEnvFlat
    (craftingworld_flat.py:40-43,57,119,185) and CraftingWorldEnvOneHot (carftingworld_onehot.py:84-103,201-203,310,
    369-371), replayed through the product's classes of the same names built with the SAME ctor kwargs the reference
    class was given (none for flat8_*: the 8x8 / 100-step defaults).  Everything each class returned is compared:
    Flat's bare frame after every reset and step; OneHot's one-hot observation / desired_goal (the un-rendered goal
    state) / init_observation; reward, done, achieved bits, step_num, ep_no, desired bits, RNG stream position."""
    import gym_craftingworld_amd as cw
    meta, kw, g = load(name)
    flat = meta['env'] == 'CraftingWorldEnvFlat'
    ck = dict(meta['ctor_kwargs'])
    if 'size' in ck:
        ck['size'] = tuple(ck['size'])
    rs = np.random.RandomState()
    rs.set_state(('MT19937', g['key0'], int(g['pos0']), 0, 0.0))
    env = (cw.CraftingWorldEnvFlat if flat else cw.CraftingWorldEnvOneHot)(**ck)
    env.np_random = rs                                   # the reference user's way of pinning a stream
    if ck.get('fixed_init_state'):                        # the pool is drawn at construction (ray.py:116-118): redo it on that stream
        env.generate_fixed_states()
    assert (env.STATE_W, env.MAX_STEPS) == (kw['size'][0], kw['max_steps'])
    if flat:
        assert env.observation_space.shape == (4 * env.STATE_W, 4 * env.STATE_W, 3)
    else:
        assert env.observation_space['desired_goal'].shape == (env.STATE_W, env.STATE_W, 12)
    ri = 0

    def check_reset(o, t):
        nonlocal ri
        assert g['r_at_step'][ri] == t
        if flat:
            assert o is env.obs_image and o.shape == (4 * env.STATE_W, 4 * env.STATE_W, 3)
            assert crc(o.astype(np.uint8)) == g['r_obs_crc'][ri], (name, 'reset frame', ri)
            assert crc(env.desired_goal.astype(np.uint8)) == g['r_desired_img_crc'][ri] and crc(env.INIT_OBS.astype(np.uint8)) == g['r_init_img_crc'][ri]
        else:
            assert o['achieved_goal'] is o['observation']
            assert crc(o['observation'].astype(np.uint8)) == g['r_obs_crc'][ri], (name, 'reset one-hot', ri)
            assert crc(o['desired_goal'].astype(np.uint8)) == g['r_desired_img_crc'][ri], (name, 'goal state', ri)
            assert crc(o['init_observation'].astype(np.uint8)) == g['r_init_img_crc'][ri]
            if ri < len(g['img_desired']):
                assert np.array_equal(o['desired_goal'], g['img_desired'][ri]) and np.array_equal(o['observation'], g['img_obs'][ri])
        bits = sum(int(b) << i for i, b in enumerate(env.desired_goal_vector[0]))
        assert bits == g['r_desired'][ri] and env.ep_no == g['r_ep_no'][ri]
        st_ = rs.get_state()                             # the caller's own RandomState IS the env's generator: it stands where the reference's stood
        assert st_[2] == g['r_rng_pos'][ri] and crc(np.asarray(st_[1], np.uint32)) == g['r_rng_crc'][ri], (name, 'rng state', ri)
        ri += 1

    check_reset(env.reset(), 0)
    for t in range(len(g['action'])):
        o, r, d, info = env.step(int(g['action'][t]))
        assert r == g['reward'][t] and d == bool(g['done'][t]), (name, t)
        assert sum(int(b) << i for i, b in enumerate(info['achieved_goal'][0])) == g['achieved'][t], (name, 'achieved', t)
        assert env.step_num == g['step_num'][t]
        ob = o if flat else o['observation']
        assert crc(ob.astype(np.uint8)) == g['obs_crc'][t], (name, 'returned observation', t)
        if d:
            check_reset(env.reset(), t + 1)
    assert ri == len(g['r_desired'])
    env.close()


def test_shards_of_self_launched_ranks_equal_the_single_batch(tmp_path):
    """Engine-level shard equivalence ACROSS PROCESSES: `bench.py --gpus 2 --shard-check T` self-launches two fresh ranks, each
    steps its contiguous env range (1 500 envs, full frames, short episodes: many resets) with actions that depend on (step, global env
    index) only and dumps per-env CRCs of its three frames, every step's reward / done, the packed state and the RNG streams; one process
    stepping all 3 000 envs does the same; the concatenated shards must equal the single batch, field by field.  Three processes use the
    GPU at most."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    common = ['--shard-check', '45', '--max-steps', '11', '--mixed-menus']
    two, one = str(tmp_path / 'two'), str(tmp_path / 'one')
    p = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--rehearse-on-one-gpu', '--dist-backend', 'gloo',
                        '--envs-per-gpu', '1500', '--shard-out', two] + common, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    assert json.loads(p.stdout.strip().splitlines()[-1])['n_gpus'] == 2
    p = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--envs-per-gpu', '3000', '--shard-out', one] + common,
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    whole = np.load(os.path.join(one, 'rank0.npz'))
    parts = [np.load(os.path.join(two, 'rank%d.npz' % r)) for r in range(2)]
    assert (int(parts[0]['lo']), int(parts[0]['hi']), int(parts[1]['lo']), int(parts[1]['hi'])) == (0, 1500, 1500, 3000)
    for key in ('obs_crc', 'goal_crc', 'init_crc', 'rng_crc', 'hdr'):
        assert np.array_equal(np.concatenate([q[key] for q in parts]), whole[key]), key
    for key in ('reward', 'done'):
        assert np.array_equal(np.concatenate([q[key] for q in parts], axis=1), whole[key]), key
    assert np.array_equal(parts[0]['counters'] + parts[1]['counters'], whole['counters'])
    assert whole['done'].sum() > 3000 * 3                # (episodes of 11 steps: every env was reset several times)


def _reference_render_of_any_state(state):
    """ray.py:442-486 restated in numpy for a caller-supplied one-hot state (test-side oracle): sum of object colours per cell,
    x4 upscale, agent = first cell with channel 8 set -> centre 2x2 white, bottom row of it in the colour of the largest hold
    channel set anywhere."""
    colors = np.array([(0, 0, 0), (110, 69, 39), (255, 105, 180), (100, 100, 200), (100, 100, 100), (0, 128, 0),
                       (205, 133, 63), (197, 91, 97), (240, 230, 140)], dtype=np.int64)          # COLORS_N, ray.py:28-30
    idx = np.where(state[:, :, 8] == 1)
    ax, ay = idx[0][0], idx[1][0]
    h, w = state.shape[:2]
    objects_n = np.concatenate((np.zeros((h, w, 1), dtype=int), state[:, :, :8]), axis=2)
    holding = np.concatenate((np.zeros((h, w, 1), dtype=int), state[:, :, 9:]), axis=2)
    img = np.tensordot(objects_n, colors, axes=1)
    img = np.repeat(np.repeat(img, 4, axis=0), 4, axis=1)
    img[ax * 4 + 1:ax * 4 + 3, ay * 4 + 1:ay * 4 + 3, :] = 255
    hold = np.max(np.argmax(holding, axis=2))
    if hold != 0:
        img[ax * 4 + 2:ax * 4 + 3, ay * 4 + 1:ay * 4 + 3] = colors[hold]
    return img


def test_render_of_arbitrary_one_hot_states():
    """render(state) accepts ANY one-hot state, as the reference's does (ray.py:442-486): several objects in one cell (their
    colours add), more than eight objects, several agent cells (the first counts), hold flags anywhere.  int64 façade: the
    reference's image exactly; uint8 façade: modulo 256; batch entry point on the device."""
    import gym_craftingworld_amd as cw
    S = 9
    rng = np.random.RandomState(12)
    exact = cw.CraftingWorldEnv(size=(S, S), reference_dtypes=True)
    wrap = cw.CraftingWorldEnv(size=(S, S))
    states = []
    for trial in range(40):
        st = (rng.rand(S, S, 12) < [0.02, 0.3, 0.08][trial % 3]).astype(int)
        st[:, :, 8] = 0
        for _ in range(1 + trial % 3):
            st[rng.randint(S), rng.randint(S), 8] = 1
        if trial % 4 == 0:
            st[:, :, 9:] = 0
        states.append(st)
        ref = _reference_render_of_any_state(st)
        got = exact.render(state=st)
        assert got.dtype == np.int64 and np.array_equal(got, ref), trial
        assert np.array_equal(wrap.render(state=st), (ref % 256).astype(np.uint8)), trial
    assert max(int(_reference_render_of_any_state(s).max()) for s in states) > 255          # sums above one byte were exercised
    batch = exact._vec.render_states(np.stack(states)).cpu().numpy().astype(np.int64)
    for i, st in enumerate(states):
        assert np.array_equal(batch[i], _reference_render_of_any_state(st)), i
    with pytest.raises(IndexError):
        exact.render(state=np.zeros((S, S, 12), int))                                         # no agent: ray.py:454 raises too
    with pytest.raises(ValueError):
        exact.render(state=np.zeros((S + 1, S, 12), int))
    exact.close(); wrap.close()


@pytest.mark.gpu
@pytest.mark.parametrize('obs_mode,consumer', [('pixels', 'reduce'), ('pixels_dirty', 'reduce'), ('pixels', 'reduce32'), ('pixels', 'conv')])
def test_policy_in_the_loop_actions_replay_through_the_oracle(obs_mode, consumer):
    """SURVEY 8b's callers: a torch policy consuming the device tensors without host copies (docs/source/envs/gen_info.rst:62-82 with a network where
    the reference samples; ray.py:376-378 hands the observation back).  Between two steps a consumer reads EVERY observation byte and produces the
    next actions from it (bench.py's --consumer), all on the env's stream, no host synchronisation inside the loop: 200 steps of 4 096 envs.  Then
    the RECORDED actions go through the oracle: rewards, dones and frames must be the oracle's, and -- `reduce` / `reduce32`, whose policy a host can
    restate exactly -- every recorded action must be what the policy computes from the ORACLE's frame of that step: a consumer that read a frame before the
    sweep (or the step kernel's repaint) had written it would have taken another action."""
    import bench
    from gym_craftingworld_amd import CraftingWorldVecEnv
    from oracle import OracleBatch
    N, T = 4096, 200
    kw = dict(size=(21, 21), max_steps=37)                # (episodes end inside the run, at spread-out steps once some succeed)
    keys, pos = _np_states(N, 52000)
    env = CraftingWorldVecEnv(N, obs_mode=obs_mode, **kw)
    env.set_rng_states(keys, pos)
    policy = bench.make_consumer(consumer, N, env.frame_shape, env.device)
    obs = env.reset()
    # spread the episode phases out: envs finish on every step, so frames of freshly reset envs are consumed on every step too
    env.set_state(step_num=(np.arange(N) % 30).astype(np.int32))
    rec_a = torch.empty((T, N), dtype=torch.uint8, device=env.device)
    rec_r = torch.empty((T, N), dtype=torch.int32, device=env.device)
    rec_d = torch.empty((T, N), dtype=torch.bool, device=env.device)
    a = policy(obs['observation'])
    for t in range(T):                                     # nothing in this loop waits for the card
        rec_a[t] = a
        obs, r, d, _ = env.step(a)
        rec_r[t] = r
        rec_d[t] = d
        a = policy(obs['observation'])
    torch.cuda.synchronize()
    acts, rews, dones = rec_a.cpu().numpy(), rec_r.cpu().numpy(), rec_d.cpu().numpy()
    final = obs['observation'].cpu().numpy()
    goal = obs['desired_goal'].cpu().numpy()
    ora = OracleBatch(N, rng_states=list(zip(keys, pos)), **kw)
    ora.reset()
    for i, e in enumerate(ora.envs):                       # the same phase spread (step_num only)
        s = e.state()
        e.set_state(s['grid'], s['init_grid'], s['agent'], s['hold'], s['achieved'], s['desired'], i % 30)
    ish = ora.envs[0].img_shape
    n_done = 0
    for t in range(T):
        if consumer in ('reduce', 'reduce32'):
            if consumer == 'reduce':
                want = np.array([int(np.ctypeslib.as_array(e.view().obs, shape=ish).sum(dtype=np.int64)) % 6 for e in ora.envs], dtype=np.uint8)
            else:                                          # the frame's bytes as little-endian int32 words, summed with int32 wrap-around, Python's modulo
                want = np.array([int(np.ctypeslib.as_array(e.view().obs, shape=ish).reshape(-1).view(np.int32).sum(dtype=np.int64)
                                     .astype(np.int32)) % 6 for e in ora.envs], dtype=np.uint8)
            bad = np.nonzero(want != acts[t])[0]
            assert bad.size == 0, ('step', t, 'envs whose action was not computed from the finished frame', bad[:8], acts[t][bad[:8]], want[bad[:8]])
        o_rew, o_done = ora.step(acts[t])
        assert np.array_equal(rews[t], o_rew), ('reward', t)
        assert np.array_equal(dones[t], o_done), ('done', t)
        n_done += int(o_done.sum())
    assert n_done > 2 * N                                  # every env was reset several times on the way
    for i in list(range(0, N, 61)) + [N - 1]:
        s = ora.envs[i].state()
        assert np.array_equal(final[i], s['obs']) and np.array_equal(goal[i], s['desired_img']), i
    assert int(env.counters[1].item()) == n_done
    env.close()


@pytest.mark.gpu
def test_facade_reference_attributes_live():
    """The attributes reference users read off CraftingWorldEnvRay (ray.py:119-141, 185-187, 624-626): None before the first reset()
    and the reference's own exceptions for a step() before it; afterwards agent_pos is a Coord-like value (.row / .col / .tuple(), == tuples),
    observation_vector holds the one-hot state and the live goal vectors, fixed_state_list the pooled placements -- those against the fixture
    the reference class itself produced with fixed_init_state=3 (every reset state of the fixture is one of the three)."""
    import gym_craftingworld_amd as cw
    from gym_craftingworld_amd.coord import GridPos
    meta, kw, g = load('ray6_fixedinit')
    env = cw.CraftingWorldEnv(**kw)
    assert env.agent_pos is None and env.obs_one_hot is None and env.INIT_OBS_VECTOR is None and env.observation_vector is None and env.observation is None
    with pytest.raises(TypeError):
        env.step(0)                                        # None + Coord, ray.py:393
    with pytest.raises(AttributeError):
        env.step(4)                                        # None.tuple(), ray.py:315
    assert env.step_num == 2                               # (counted before the failure, ray.py:309)
    env.step_num = 0
    with pytest.raises(IndexError):
        env.step(6)
    assert [getattr(a, 'name', a) for a in env.ACTIONS] == ['up', 'right', 'down', 'left', 'pickup', 'drop'] and env.ACTIONS[0].tuple() == (-1, 0)
    env.set_rng_state(g['key0'], int(g['pos0']))
    pool = env.generate_fixed_states(3)                    # the constructor's draw, redone on the injected stream (ray.py:116-118)
    assert len(pool) == 3 and all(p.shape == (6, 6, 12) and p.sum() == 9 and p[:, :, 9:].sum() == 0 for p in pool)
    with pytest.raises(ValueError):
        env.generate_fixed_states(4)

    def key_of(grid, agent):
        return (np.asarray(grid, np.uint8).tobytes(), tuple(int(x) for x in agent))
    mine = {key_of((p[:, :, :8] * np.arange(1, 9)).sum(2), np.argwhere(p[:, :, 8] == 1)[0]) for p in env.fixed_state_list}
    theirs = {key_of(g['r_grid'][i], g['r_agent'][i]) for i in range(len(g['r_grid']))}
    assert len(mine) == 3 and theirs <= mine and len(theirs) >= 2
    obs = env.reset()
    ap = env.agent_pos
    assert isinstance(ap, GridPos) and ap.tuple() == tuple(g['r_agent'][0]) and ap == tuple(g['r_agent'][0]) and (ap.row, ap.col) == tuple(g['r_agent'][0])
    assert ap.max_row == 5 and ap.max_col == 5 and (ap + env.ACTIONS[0]).row == max(ap.row - 1, 0)
    r, c = ap
    ov = env.observation_vector
    assert set(ov) == {'observation', 'desired_goal', 'achieved_goal', 'init_observation'}
    assert ov['observation'].shape == (6, 6, 12) and ov['observation'][r, c, 8] == 1 and np.array_equal(ov['observation'], ov['init_observation'])
    assert ov['desired_goal'] is env.desired_goal_vector and ov['achieved_goal'] is env.achieved_goal_vector and ov['desired_goal'].shape == (1, 9)
    assert np.array_equal((ov['observation'][:, :, :8] * np.arange(1, 9)).sum(2), g['r_grid'][0])
    for t in range(60):
        o, rew, d, info = env.step(int(g['action'][t]))
        assert rew == g['reward'][t] and env.agent_pos == tuple(g['agent'][t]), t
        if d:
            env.reset()
    no_pool = cw.CraftingWorldEnv(size=(5, 5))
    with pytest.raises(AttributeError):
        no_pool.fixed_state_list
    no_pool.close()
    env.close()


@pytest.mark.gpu
def test_synchronous_calls_wait_for_the_engines_own_streams_only():
    """cw_seed_* / cw_get_mt / cw_get_state / checkpoints no longer synchronise the DEVICE (round 4: hipDeviceSynchronize stalled every other engine
    and any learner on the card): they wait for the streams this engine was handed, and copy on a stream of their own.  Work enqueued on a torch
    SIDE stream (non-blocking: the null stream does not wait for it) must be complete in what they return, with no synchronisation by the caller."""
    from gym_craftingworld_amd import CraftingWorldVecEnv
    N, T = 20000, 120
    kw = dict(size=(21, 21), max_steps=25, obs_mode='pixels')
    a, b = CraftingWorldVecEnv(N, seed=4, **kw), CraftingWorldVecEnv(N, seed=4, **kw)
    gen = torch.Generator(device='cuda').manual_seed(3)
    acts = torch.randint(0, 6, (T, N), device='cuda', dtype=torch.uint8, generator=gen)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    a._settle = lambda: None                             # (the Python layer's own wait for torch's current stream: off, this is about the library's)
    with torch.cuda.stream(side):
        a.reset()
        for t in range(T):
            a.step_async(acts[t])
        sa = a.get_state()                               # no wait in between: T sweeps are still queued on `side`
        ka, pa = a.get_rng_states()
    b.reset()
    for t in range(T):
        b.step_async(acts[t])
    torch.cuda.synchronize()
    sb = b.get_state()
    kb, pb = b.get_rng_states()
    for k in sa:
        assert np.array_equal(sa[k], sb[k]), k
    assert np.array_equal(pa, pb) and np.array_equal(ka[:, 1:], kb[:, 1:])
    with torch.cuda.stream(side):                        # ... and a re-seed issued behind queued work takes effect after it, not in the middle of it
        for t in range(30):
            a.step_async(acts[t])
        a.seed(77)
        a.reset()
    for t in range(30):
        b.step_async(acts[t])
    b.seed(77)
    b.reset()
    torch.cuda.synchronize()
    assert torch.equal(a.hdr, b.hdr) and torch.equal(a._obs, b._obs) and torch.equal(a.counters, b.counters)
    a.close(); b.close()


@pytest.mark.gpu
def test_synchronous_calls_after_a_graph_replay_wait_for_the_device():
    """A HIP graph captured from cw_step_many replays on whatever stream the caller launches it on -- one the engine was never handed.  Since round 5 the
    synchronous entry points wait for the engine's OWN streams only; an engine whose steps have ever been captured therefore falls back to a device-wide
    wait (a sticky flag, set at capture): get_state / get_rng_states / save_checkpoint right behind replays on a side stream, with the Python layer's own
    wait switched off, must see every replayed step."""
    from gym_craftingworld_amd import CraftingWorldVecEnv
    N, K, R = 30000, 16, 40
    kw = dict(size=(21, 21), max_steps=23, obs_mode='pixels', seed=9)
    a, b = CraftingWorldVecEnv(N, **kw), CraftingWorldVecEnv(N, **kw)
    acts = torch.randint(0, 6, (K, N), device='cuda', dtype=torch.uint8, generator=torch.Generator(device='cuda').manual_seed(4))
    a.reset(), b.reset()
    graph = a.capture_steps(acts)
    torch.cuda.synchronize()
    a._settle = lambda: None                             # (only the library's own wait is under test)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(R):
            graph.replay()                               # R x K sweeps of 30 000 frames queued on `side`: ~60 ms of work
    sa = a.get_state()                                   # ... and no wait by the caller
    ka, pa = a.get_rng_states()
    for _ in range(R):
        b.step_many(acts)
    torch.cuda.synchronize()
    sb = b.get_state()
    kb, pb = b.get_rng_states()
    assert int(sa['step_num'].max()) > 0
    for k in sa:
        assert np.array_equal(sa[k], sb[k]), k
    assert np.array_equal(pa, pb) and np.array_equal(ka, kb)
    assert torch.equal(a._obs, b._obs) and torch.equal(a.counters, b.counters)
    a.close(); b.close()


@pytest.mark.gpu
@pytest.mark.parametrize('obs_mode', ['state', 'pixels_dirty'])
def test_replayed_graph_gets_its_records_back_after_a_reseed(obs_mode):
    """A captured graph bakes cw_refill_kernel in with all_envs = 0: it refills the LIST.  A re-seed drops every look-ahead record (they were computed from
    the old streams) and empties the list; finished envs then reset the slow way -- and (round 5) put themselves on the list, so the next replay's refill
    gives them records again.  Round 4 never did: every later episode of such an engine was reset the slow way, equal results at a permanent cost.
    counters[5] (engine-private) counts slow resets: it must stop growing once the first episodes after the re-seed are over.  Results equal an
    engine stepped eagerly all along."""
    from gym_craftingworld_amd import CraftingWorldVecEnv
    N, K = 6000, 6
    kw = dict(size=(9, 9), max_steps=11, obs_mode=obs_mode, seed=5)
    g_env, e_env = CraftingWorldVecEnv(N, **kw), CraftingWorldVecEnv(N, **kw)
    gen = torch.Generator(device='cuda').manual_seed(8)
    ring = torch.randint(0, 6, (K, N), device='cuda', dtype=torch.uint8, generator=gen)
    g_env.reset(); e_env.reset()
    graph = g_env.capture_steps(ring)
    assert int(g_env.counters[0]) == 0                   # capturing took no step (round 4's eager warm-up advanced every env by K steps)
    for _ in range(4):
        graph.replay()
        e_env.step_many(ring)
    g_env.seed(123); e_env.seed(123)                     # every record dropped on both
    slow = []
    for rep in range(14):
        graph.replay()
        e_env.step_many(ring)
        torch.cuda.synchronize()
        slow.append(int(g_env._counters_raw[5]))
    assert slow[1] >= N                                  # the first episodes after the re-seed ended without a record (every env once) ...
    assert slow[-1] - slow[2] < N // 50, slow            # ... and from then on finished envs find one again: 12 replays = 6 more episodes of every env (round 4:
    #                                                      6 x N more slow resets), but for the few that finish twice between two refills
    assert torch.equal(g_env.hdr, e_env.hdr) and torch.equal(g_env.slot_pos, e_env.slot_pos) and torch.equal(g_env.counters, e_env.counters)
    ka, pa = g_env.get_rng_states(); kb, pb = e_env.get_rng_states()
    assert np.array_equal(pa, pb) and np.array_equal(ka[:, 1:], kb[:, 1:])
    g_env.close(); e_env.close()


@pytest.mark.gpu
@pytest.mark.parametrize('obs_mode', ['state', 'pixels'])
def test_checkpoints_cross_lookahead_settings(obs_mode, tmp_path, monkeypatch):
    """Look-ahead records are work done ahead, not state: a checkpoint written by an engine that keeps them resumes on one that keeps none
    (CW_TUNE_LOOKAHEAD=0; the streams of envs whose record waited are rewound by the record's draws, cwh_mt_rewind) and the other way round (the
    records are recomputed at the next refill) -- either way bit-identical to the run that never stopped, RNG streams included."""
    from gym_craftingworld_amd import CraftingWorldVecEnv
    N = 700
    kw = dict(size=(8, 8), max_steps=13, obs_mode=obs_mode, seed=3)
    gen = torch.Generator(device='cuda').manual_seed(16)
    acts = torch.randint(0, 6, (150, N), device='cuda', dtype=torch.uint8, generator=gen)
    a = CraftingWorldVecEnv(N, **kw)                      # keeps records
    assert a.tuner_state()['lookahead'] == 1
    a.reset()
    for t in range(37):
        a.step(acts[t])
    pa = str(tmp_path / 'with_records')
    a.save_checkpoint(pa)
    monkeypatch.setenv('CW_TUNE_LOOKAHEAD', '0')
    b = CraftingWorldVecEnv(N, **dict(kw, seed=99))       # keeps none
    monkeypatch.delenv('CW_TUNE_LOOKAHEAD')
    assert b.tuner_state()['lookahead'] == 0
    b.load_checkpoint(pa)
    ka, qa = a.get_rng_states(); kb, qb = b.get_rng_states()
    assert np.array_equal(qa % 624, qb % 624) and np.array_equal(ka[:, 1:], kb[:, 1:])
    for t in range(37, 90):
        _, ra, da, _ = a.step(acts[t])
        _, rb, db, _ = b.step(acts[t])
        assert torch.equal(ra, rb) and torch.equal(da, db), t
    assert torch.equal(a.hdr, b.hdr) and torch.equal(a.slot_pos, b.slot_pos) and torch.equal(a.counters, b.counters)
    pb = str(tmp_path / 'without_records')
    b.save_checkpoint(pb)
    assert os.path.getsize(pb) < os.path.getsize(pa)
    c = CraftingWorldVecEnv(N, **dict(kw, seed=5))        # keeps records again
    c.load_checkpoint(pb)
    for t in range(90, 150):
        oa, ra, da, _ = a.step(acts[t])
        oc, rc, dc, _ = c.step(acts[t])
        assert torch.equal(ra, rc) and torch.equal(da, dc), t
    assert torch.equal(a.hdr, c.hdr) and torch.equal(a.slot_pos, c.slot_pos) and torch.equal(a.counters, c.counters)
    if obs_mode == 'pixels':
        for k in oa:
            assert torch.equal(oa[k], oc[k]), k
    ka, qa = a.get_rng_states(); kc, qc = c.get_rng_states()
    assert np.array_equal(qa % 624, qc % 624) and np.array_equal(ka[:, 1:], kc[:, 1:])
    assert int(c._counters_raw[5]) < N // 4              # (c found records again after its first refill: no slow resets to speak of)
    for e in (a, b, c):
        e.close()
