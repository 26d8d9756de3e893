"""CPU-only tests of the product's host logic: the C-ABI library loads and exports every symbol
include/craftingworld.h declares, MT19937 state conversion (numpy form <-> the engine's
consume-and-replace form) is stream-exact, seeding helpers, and the product never touches the
oracle.  No compute entry point is called (no GPU here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib():
    import __graft_entry__ as g
    g.build()
    from gym_craftingworld_amd import _lib
    return _lib.load()


def test_header_symbols_exported(lib):
    hdr = open(os.path.join(ROOT, 'include', 'craftingworld.h')).read()
    declared = set(re.findall(r'^\s*(?:const\s+)?(?:int|size_t|char\s*\*|const char \*)\s*\*?\s*(cw_[a-z_0-9]+)\s*\(', hdr, re.M))
    assert len(declared) >= 16, declared
    from gym_craftingworld_amd import _lib
    assert declared == set(_lib.ABI), (declared ^ set(_lib.ABI))
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.cw_abi_version() == _lib.CW_ABI_VERSION == 5


def test_struct_layouts_match_header(lib, tmp_path):
    """The header is plain C (compiles with gcc -std=c99 -pedantic) and the ctypes mirrors in _lib.py
    have the sizes and field offsets the C compiler gives the header's structs."""
    import subprocess
    from gym_craftingworld_amd import _lib
    structs = {'cw_task_menu': _lib.cw_task_menu, 'cw_config': _lib.cw_config, 'cw_buffer_table': _lib.cw_buffer_table,
               'cw_state_view': _lib.cw_state_view, 'cw_profile': _lib.cw_profile, 'cw_tuner_state': _lib.cw_tuner_state}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "craftingworld.h"', 'int main(void){']
    for name, st in structs.items():
        lines.append('printf("%s %%zu\\n", sizeof(%s));' % (name, name))
        for f, _ in st._fields_:
            lines.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (name, f, name, f))
    lines.append('return 0;}')
    src = tmp_path / 'layout.c'
    src.write_text('\n'.join(lines))
    exe = tmp_path / 'layout'
    subprocess.check_call(['gcc', '-std=c99', '-pedantic', '-Wall', '-Werror', '-I', os.path.join(ROOT, 'include'),
                           str(src), '-o', str(exe)])
    got = dict(l.split() for l in subprocess.check_output([str(exe)]).decode().splitlines())
    for name, st in structs.items():
        assert int(got[name]) == C.sizeof(st), name
        for f, _ in st._fields_:
            assert int(got['%s.%s' % (name, f)]) == getattr(st, f).offset, (name, f)


def _engine_next(s, k):
    """cw_mt.h CwMt::next() restated for the test (consume-and-replace)."""
    cur, nxt, far = int(s[k]), int(s[(k + 1) % 624]), int(s[(k + 397) % 624])
    y = (cur & 0x80000000) | (nxt & 0x7fffffff)
    s[k] = far ^ (y >> 1) ^ (0x9908b0df if y & 1 else 0)
    o = cur
    o ^= o >> 11
    o ^= (o << 7) & 0x9d2c5680
    o ^= (o << 15) & 0xefc60000
    o ^= o >> 18
    return o & 0xFFFFFFFF, (k + 1) % 624


@pytest.mark.parametrize('seed,burn', [(0, 0), (1, 5), (7, 623), (123, 624), (99, 1000), (5, 227), (6, 397)])
def test_mt_conversion_roundtrip(lib, seed, burn):
    rs = np.random.RandomState(seed)
    if burn:
        rs.randint(0, 2**32, size=burn, dtype=np.uint32)
    st = rs.get_state()
    s = st[1].astype(np.uint32).copy()
    idx = lib.cwh_mt_from_numpy(s.ctypes.data_as(C.c_void_p), int(st[2]))
    draws = 1500
    ref = rs.randint(0, 2**32, size=draws, dtype=np.uint32)
    got = np.empty(draws, dtype=np.uint32)
    k = idx
    for i in range(draws):
        got[i], k = _engine_next(s, k)
    assert np.array_equal(ref, got)
    # export back: numpy continues identically from the exported state
    key = np.empty(624, dtype=np.uint32)
    lib.cwh_mt_to_numpy(s.ctypes.data_as(C.c_void_p), k, key.ctypes.data_as(C.c_void_p))
    rs2 = np.random.RandomState()
    rs2.set_state(('MT19937', key, k, 0, 0.0))
    assert np.array_equal(rs.randint(0, 2**32, size=2000, dtype=np.uint32),
                          rs2.randint(0, 2**32, size=2000, dtype=np.uint32))
    # and, except for the unrecoverable (and unused) low 31 bits of key[0], it IS numpy's key
    st2 = rs_state_at(seed, burn + draws)
    if k != 0:
        assert st2[2] == k
        assert np.array_equal(st2[1][1:], key[1:]) and (int(st2[1][0]) ^ int(key[0])) & 0x80000000 == 0


def rs_state_at(seed, n):
    rs = np.random.RandomState(seed)
    if n:
        rs.randint(0, 2**32, size=n, dtype=np.uint32)
    return rs.get_state()


def test_init_genrand_matches_numpy(lib):
    for seed in (0, 1, 42, 2**32 - 1):
        s = np.empty(624, dtype=np.uint32)
        lib.cwh_mt_init_genrand(s.ctypes.data_as(C.c_void_p), seed)
        assert np.array_equal(s, np.random.RandomState(seed).get_state()[1])


def test_seeding_gym021_shape():
    from gym_craftingworld_amd import seeding
    rs, seed = seeding.np_random(123)
    assert seed == 123 and isinstance(rs, np.random.RandomState)
    key, pos = seeding.mt_state_from_seed(123)
    assert key.shape == (624,) and pos == 624
    a, _ = seeding.np_random(123)
    assert a.randint(1 << 30) == rs.randint(1 << 30)
    with pytest.raises(ValueError):
        seeding.create_seed(-1)


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    import gym_craftingworld_amd as g
    with pytest.raises(g.CraftingWorldError):
        g.CraftingWorldVecEnv(2)
    with pytest.raises(ValueError):
        g.CraftingWorldVecEnv(2, size=(6, 4))


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under gym_craftingworld_amd/ may reference it."""
    pkg = os.path.join(ROOT, 'gym_craftingworld_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.cpp', '.hip', '.h')):
                txt = open(os.path.join(dirpath, f), errors='replace').read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', txt, re.M), f
                assert 'cw_oracle' not in txt and 'libcw_oracle' not in txt, f


# ------------------------------------------------------------------ property tests (hypothesis)
from hypothesis import given, settings, strategies as st  # noqa: E402


@settings(max_examples=60, deadline=None)
@given(seed=st.integers(0, 2**32 - 1), burn=st.integers(0, 1400), draws=st.integers(1, 900))
def test_mt_conversion_property(seed, burn, draws):
    """For any numpy RandomState state: to engine form, consume `draws` words with the engine's
    recurrence, export back -> numpy continues with the identical stream (and the raw words match)."""
    from hostlib import host_lib
    lib = host_lib()[1]
    rs = np.random.RandomState(seed)
    if burn:
        rs.randint(0, 2**32, size=burn, dtype=np.uint32)
    state = rs.get_state()
    s = state[1].astype(np.uint32).copy()
    k = lib.cwh_mt_from_numpy(s.ctypes.data_as(C.c_void_p), int(state[2]))
    ref = rs.randint(0, 2**32, size=draws, dtype=np.uint32)
    got = np.empty(draws, dtype=np.uint32)
    for i in range(draws):
        got[i], k = _engine_next(s, k)
    assert np.array_equal(ref, got)
    key = np.empty(624, dtype=np.uint32)
    lib.cwh_mt_to_numpy(s.ctypes.data_as(C.c_void_p), k, key.ctypes.data_as(C.c_void_p))
    rs2 = np.random.RandomState()
    rs2.set_state(('MT19937', key, k, 0, 0.0))
    assert np.array_equal(rs.randint(0, 2**32, size=700, dtype=np.uint32), rs2.randint(0, 2**32, size=700, dtype=np.uint32))


@settings(max_examples=80, deadline=None)
@given(seed=st.integers(0, 2**32 - 1), burn=st.integers(0, 1400), ahead=st.integers(0, 3000))
def test_mt_rewind_property(seed, burn, ahead):
    """Look-ahead records leave an env's stream one reset AHEAD; cw_get_mt reports the position before it by rewinding the exported numpy state
    by the record's draws (cwh_mt_rewind: within the generation, and through cwh_mt_untwist across any number of generations).  For any state:
    export `ahead` draws later (through the engine form, as cw_get_mt does), rewind by `ahead` -> the stream from the state `burn` draws in,
    the same position modulo 624 and the same key words from there on."""
    from hostlib import host_lib
    lib = host_lib()[1]
    rs = np.random.RandomState(seed)
    if burn:
        rs.randint(0, 2**32, size=burn, dtype=np.uint32)
    want = rs.get_state()
    state = rs.get_state()
    s = state[1].astype(np.uint32).copy()
    k = lib.cwh_mt_from_numpy(s.ctypes.data_as(C.c_void_p), int(state[2]))
    for _ in range(ahead):
        _, k = _engine_next(s, k)
    key = np.empty(624, dtype=np.uint32)
    lib.cwh_mt_to_numpy(s.ctypes.data_as(C.c_void_p), k, key.ctypes.data_as(C.c_void_p))
    pos = C.c_int32(k)
    lib.cwh_mt_rewind(key.ctypes.data_as(C.c_void_p), C.byref(pos), ahead)
    assert 0 <= pos.value <= 624 and pos.value % 624 == int(want[2]) % 624
    rs2 = np.random.RandomState()
    rs2.set_state(('MT19937', key, pos.value, 0, 0.0))
    assert np.array_equal(rs.randint(0, 2**32, size=1500, dtype=np.uint32), rs2.randint(0, 2**32, size=1500, dtype=np.uint32))
    if int(want[2]) < 624:          # numpy holds the same generation: the words not yet consumed are its own
        assert np.array_equal(np.asarray(want[1])[max(pos.value, 1):], key[max(pos.value, 1):])


@settings(max_examples=200, deadline=None)
@given(world=st.integers(1, 64), total=st.integers(1, 2**22))
def test_shard_range_property(world, total):
    from gym_craftingworld_amd.sharding import shard_range
    r = [shard_range(g, world, total) for g in range(world)]
    assert r[0][0] == 0 and r[-1][1] == total
    assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
    sizes = [b - a for a, b in r]
    assert max(sizes) - min(sizes) <= 1 and min(sizes) >= 0


def test_bench_cpu_baseline_leg_reports_port_and_calibration():
    """bench.py's cpu_baseline object (task contract (4)): kind "port", the cores used, a described sample, and
    the committed build-container calibration against the reference's own Python."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('cw_bench', os.path.join(root, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    out = bench.cpu_baseline(5, 20, seconds=0.2)
    assert out['kind'] == 'port' and out['unit'] == 'env-steps/s' and out['cores'] >= 1 and out['value'] > 0
    assert 'envs x' in out['sample'] and out['dirty_cell_value'] > 0
    assert 'whole frame' in out['like_for_like']           # `value` is like for like with the GPU headline
    c = out['calibration']
    assert c is not None and 20 < c['port_over_reference_1core'] < 2000
    assert abs(c['reference_equivalent_env_steps_per_s'] * c['port_over_reference_1core'] - out['dirty_cell_value']) < 1e-6 * out['dirty_cell_value']


def test_vector_env_surface_matches_gym_vector():
    """SURVEY 8b "interface": the gym.vector.VectorEnv surface of the batch class -- attribute and method names with the
    argument lists gym.vector (<= 0.21) gives them -- checked without constructing an engine (no GPU here)."""
    import inspect
    from gym_craftingworld_amd import CraftingWorldVecEnv as V
    sig = lambda f: list(inspect.signature(f).parameters)  # noqa: E731
    assert V.is_vector_env is True and V.viewer is None
    assert sig(V.seed) == ['self', 'seed']                       # seed(seeds=None): int | list | None
    assert sig(V.reset_async) == ['self'] and sig(V.reset_wait) == ['self'] and sig(V.reset) == ['self']
    assert sig(V.step_async) == ['self', 'actions'] and sig(V.step_wait) == ['self'] and sig(V.step) == ['self', 'actions']
    assert sig(V.close_extras) == ['self', 'kwargs'] and sig(V.close) == ['self', 'kwargs']
    assert isinstance(V.closed, property) and isinstance(V.unwrapped, property)
    ctor = sig(V.__init__)
    for kw in ('num_envs', 'size', 'fixed_init_state', 'max_steps', 'store_gif', 'render_save_rate', 'task_list',
               'selected_tasks', 'number_of_tasks', 'stacking', 'reward_style'):      # ray.py:59-60 + num_envs
        assert kw in ctor, kw
    src = inspect.getsource(V.__init__)
    for attr in ('self.num_envs', 'self.single_observation_space', 'self.single_action_space', 'self.observation_space',
                 'self.action_space'):
        assert attr in src, attr
    v = object.__new__(V)                                        # no engine: closed, and close() is a no-op
    assert v.closed is True
    v.close()


def test_batch_space_adds_a_leading_axis():
    from gym_craftingworld_amd.spaces import Box, Dict, Discrete, batch_space
    single = Dict({'observation': Box(0, 255, (84, 84, 3), np.uint8), 'hdr': Box(0, 255, (16,), np.uint8),
                   'slot_pos': Box(-2, 32767, (8,), np.int16)})
    b = batch_space(single, 7)
    assert b['observation'].shape == (7, 84, 84, 3) and b['observation'].dtype == np.uint8
    assert b['hdr'].shape == (7, 16) and b['slot_pos'].shape == (7, 8) and b['slot_pos'].dtype == np.int16
    assert int(b['slot_pos'].low.min()) == -2 and int(b['slot_pos'].high.max()) == 32767 and int(b['observation'].high.max()) == 255
    assert batch_space(Discrete(6), 5).nvec.tolist() == [6] * 5




def test_gridpos_keeps_the_reference_coord_contract():
    """agent_pos / ACTIONS of the N=1 classes are GridPos values (gym_craftingworld_amd/coord.py), the counterpart of the reference's Coord
    (coordinates.py:6-42): row / col / max_row / max_col / name, + and - clamped to the grid (coordinates.py:22-30), tuple(), str() -- and, beyond the
    reference, equality with a plain (row, col) pair and unpacking like one (what this attribute was in earlier rounds)."""
    from gym_craftingworld_amd.coord import Coord, GridPos
    assert Coord is GridPos
    p = GridPos(3, 0, 20, 20)
    assert (p.row, p.col, p.max_row, p.max_col, p.name) == (3, 0, 20, 20, None)
    up, right, down, left = GridPos(-1, 0, name='up'), GridPos(0, 1, name='right'), GridPos(1, 0, name='down'), GridPos(0, -1, name='left')
    assert (p + up).tuple() == (2, 0) and (p + left).tuple() == (3, 0) and (p + left) == p          # the west wall: "unchanged pos => fail", ray.py:395-396
    q = GridPos(20, 20, 20, 20)
    assert (q + down) == q and (q + right) == q and (q - down).tuple() == (19, 20) and (q + up).max_row == 20
    assert (GridPos(0, 0, 4, 4) - GridPos(1, 1)).tuple() == (0, 0)
    assert p == (3, 0) and p == [3, 0] and p == GridPos(3, 0) and p != (3, 1) and p != GridPos(0, 3) and not (p == 'up') and p != None  # noqa: E711
    r, c = p
    assert (r, c) == (3, 0) and p[0] == 3 and p[1] == 0 and len(p) == 2 and tuple(p) == p.tuple()
    assert str(p) == '(3, 0)' and hash(p) == hash((3, 0)) and {p: 1}[GridPos(3, 0)] == 1
    assert up.name == 'up' and 'up' in repr(up)


def test_facades_declare_the_reference_attributes():
    """Every public attribute SURVEY 8b lists for CraftingWorldEnvRay (ray.py:75-141), plus observation_vector (ray.py:185-187), fixed_state_list
    (ray.py:116-118) and generate_fixed_states (ray.py:149-154), exists on the N=1 classes -- as a property, a method or an instance attribute the
    constructor sets (read from the source: no GPU here)."""
    import inspect
    import gym_craftingworld_amd.env as envmod
    src = inspect.getsource(envmod.CraftingWorldEnv)
    names = ['observation_space', 'action_space', 'np_random', 'obs_one_hot', 'agent_pos', 'desired_goal_vector', 'achieved_goal_vector',
             'INIT_OBS_VECTOR', 'step_num', 'ep_no', 'MAX_STEPS', 'STATE_W', 'STATE_H', 'observation_vector', 'observation_vector_space',
             'fixed_state_list', 'generate_fixed_states', 'obs_image', 'INIT_OBS', 'desired_goal', 'observation', 'ACTIONS', 'task_list',
             'selected_tasks', 'number_of_tasks', 'stacking', 'fixed_init_state', 'store_gif', 'render_save_rate', 'seed', 'reset', 'step', 'render',
             'compute_reward', 'allow_gif_storage', 'metadata', 'reward_range', 'spec', 'unwrapped', 'close', '__enter__', '__exit__', 'compute_reward_equal', 'compute_reward_subset', 'short_circuit_check', 'one_hot', 'translate_one_hot']
    for n in names:
        assert hasattr(envmod.CraftingWorldEnv, n) or ('self.%s = ' % n) in src or ('self.%s, ' % n) in src or (', self.%s = ' % n) in src, n
    for cls in (envmod.CraftingWorldEnvFlat, envmod.CraftingWorldEnvOneHot, envmod.CraftingWorldEnvAltObs):
        assert issubclass(cls, envmod.CraftingWorldEnv)
    import gym_craftingworld_amd as cw
    import gym_craftingworld_amd.envs as refpath                         # the reference's import path for its four classes (envs/__init__.py:1-4)
    assert refpath.CraftingWorldEnvRay is cw.CraftingWorldEnvRay is envmod.CraftingWorldEnv and refpath.CraftingWorldEnvFlat is envmod.CraftingWorldEnvFlat
    assert refpath.CraftingWorldEnvOneHot is envmod.CraftingWorldEnvOneHot and refpath.CraftingWorldEnvAltObs is envmod.CraftingWorldEnvAltObs
    for prop in ('agent_pos', 'obs_one_hot', 'INIT_OBS_VECTOR', 'observation_vector', 'fixed_state_list', 'np_random'):
        assert isinstance(getattr(envmod.CraftingWorldEnv, prop), property), prop


def test_facade_helper_methods_follow_the_reference_rules():
    """The reference class's small public helpers (ray.py:747-767, 784-799) on the N=1 facade, without an engine (no GPU here): both reward rules on
    every pair of 9-bit goal vectors that differs from equality in an interesting way, the chunked comparison (= plain equality), a cell's one-hot
    row and its inverse."""
    import itertools
    import gym_craftingworld_amd.env as envmod
    E = envmod.CraftingWorldEnv
    env = E.__new__(E)                                   # the helpers only read MAX_STEPS
    env.MAX_STEPS = 300
    rng = np.random.RandomState(0)
    vecs = [np.zeros(9, int), np.ones(9, int)] + [rng.randint(0, 2, 9) for _ in range(40)]
    for a, d in itertools.product(vecs, vecs):
        assert env.compute_reward_equal(a, d) == (300 if (a == d).all() else -1)
        assert env.compute_reward_subset(a, d) == (300 if np.max(d - a) == 0 else -1)
        assert E.short_circuit_check(d, a, 4) == bool((a == d).all())
    # (1, 9)-shaped live vectors, as info['achieved_goal'] / info['desired_goal'] hand them out (ray.py:376-378)
    assert env.compute_reward_equal(vecs[5][None], vecs[5][None]) == 300 and env.compute_reward_subset(vecs[1][None], vecs[0][None]) == -1
    assert env.compute_reward_subset(np.ones(9, int), np.array([1, 0, 0, 0, 0, 0, 0, 0, 0])) == 300      # desired is a subset of achieved
    assert env.compute_reward_subset(np.zeros(9, int), np.array([1, 0, 0, 0, 0, 0, 0, 0, 0])) == -1
    assert env.one_hot() == [0] * 12 and env.one_hot(obj=3) == [0, 0, 0, 1] + [0] * 8
    assert env.one_hot(agent=True, holding=2) == [0] * 8 + [1, 0, 0, 1] and env.one_hot(obj=7, agent=True, holding=0) == [0] * 7 + [1, 1, 1, 0, 0]
    for obj, agent, holding in itertools.product([None] + list(range(8)), [False, True], [None, 0, 1, 2]):
        o, ag, h = E.translate_one_hot(np.array(env.one_hot(obj, agent, holding)))
        assert (o, bool(ag), h) == (obj, agent, holding)


def test_module_constants_are_the_reference_palette_and_ids():
    """The package's module-level constants (ray.py:15-46: action ids, OBJECTS, the palette, the default grid) against the oracle's render: an object's
    tile is COLORS[k] = COLORS_N[k + 1], the floor COLORS_N[0], the agent's centre white with the held item's colour on the lower row = 255 - COLORS_H."""
    import gym_craftingworld_amd as cw
    from oracle.oracle import render
    assert (cw.UP, cw.RIGHT, cw.DOWN, cw.LEFT, cw.PICKUP, cw.DROP) == (0, 1, 2, 3, 4, 5) and cw.ACTION_NAMES[cw.DOWN] == 'down'
    assert (cw.STATE_W, cw.STATE_H, cw.MAX_STEPS) == (21, 21, 300) and len(cw.COLORS) == len(cw.OBJECTS) == 8 and cw.PICKUPABLE == cw.OBJECTS[:3]
    S = 5
    for k in range(8):
        g = np.zeros((S, S), np.uint8)
        g[1, 2] = k + 1
        img = render(S, g, (4, 4), 0)
        assert tuple(int(x) for x in img[4, 8]) == cw.COLORS[k] == cw.COLORS_N[k + 1] and tuple(int(x) for x in img[0, 0]) == cw.COLORS_N[0]
    for h in (1, 2, 3):
        img = render(S, np.zeros((S, S), np.uint8), (2, 2), h)
        assert tuple(int(x) for x in img[9, 9]) == (255, 255, 255)
        assert tuple(255 - int(x) for x in img[10, 9]) == cw.COLORS_H[h - 1] and tuple(int(x) for x in img[10, 10]) == cw.COLORS[h - 1]


def test_docs_name_only_kernels_that_exist():
    """The public header, bench.py and the package's docstrings name kernels a rocprofv3 trace of a run would list: every cw_*_kernel they
    mention must be a __global__ of csrc/cw_kernels.hip (round 4 left names of kernels that were gone)."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hip = open(os.path.join(root, 'gym_craftingworld_amd', 'csrc', 'cw_kernels.hip')).read()
    defined = set(re.findall(r'__global__[^;{]*?\b(cw_\w+_kernel)\s*\(', hip))
    assert {'cw_step_fused_kernel', 'cw_render_pieces_kernel', 'cw_refill_kernel', 'cw_reset_kernel'} <= defined
    defined.add('cw_sweep_kernel')                       # (a host function of cw_kernels.hip: picks the sweep's template instance)
    files = ['include/craftingworld.h', 'bench.py', 'INTEGRATION.md', 'README.md', 'gym_craftingworld_amd/vec_env.py', 'gym_craftingworld_amd/env.py',
             'gym_craftingworld_amd/csrc/cw_engine.cpp', 'gym_craftingworld_amd/csrc/cw_layout.h', 'gym_craftingworld_amd/csrc/cw_kernels.hip']
    for f in files:
        for k in set(re.findall(r'\bcw_\w+_kernel\b', open(os.path.join(root, f)).read())):
            assert k in defined, (f, k)
