"""Load the golden fixtures captured from the reference (tools/gen_golden.py) and replay them."""
import glob
import json
import os
import zlib

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def fixture_names():
    """the trajectory fixtures (the *_alias ones hold an op script instead: run_alias_script below)"""
    names = sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN, '*.npz')))
    return [n for n in names if not n.endswith('_alias')]


def load(name):
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    d = {k: z[k] for k in z.files}
    meta = json.loads(bytes(d.pop('meta')).decode())
    kw = meta['kwargs']
    kw['size'] = tuple(kw['size'])
    meta['stacked_obs'] = bool(kw.pop('stacked_obs', False))      # (a kwarg of the N=1 AltObs class only: the return convention, not the dynamics)
    return meta, kw, d


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


def one_hot(grid, agent, hold=0):
    """(S,S) cell codes + agent (r,c) + hold -> the reference's (S,S,12) one-hot state (ray.py:94-98: channels 0-7
    objects, 8 agent, 9-11 the held sticks/axe/hammer at the agent's cell) as uint8."""
    g = np.asarray(grid)
    oh = np.zeros(g.shape + (12,), dtype=np.uint8)
    r, c = np.nonzero(g)
    oh[r, c, g[r, c] - 1] = 1
    oh[agent[0], agent[1], 8] = 1
    if hold:
        oh[agent[0], agent[1], 8 + int(hold)] = 1
    return oh


# ---------------------------------------------------------------------------------------------------------------------------------------
# The ALIASING script (tests/golden/ray5_alias.npz): a list of (op, arg) run through an env of the reference's gym.Env shape -- the reference's own
# CraftingWorldEnvRay when tools/gen_golden.py captures the fixture, this package's CraftingWorldEnv when the tests replay it -- that exercises what
# a caller sees of the env's OBJECTS rather than of its numbers: np_random as the live generator (ray.py:145-147: draws from it, seed / set_state on
# it, an assigned RandomState), the goal vectors rebound by reset() (ray.py:170, 176: a kept terminal `info` survives), negative action ids
# (ACTIONS[action], ray.py:308).  run_alias_script returns one row of up to 8 integers per op; the fixture holds the reference's rows.
A_RESET, A_STEP, A_RANDINT, A_SHUFFLE, A_RSEED, A_RSETSTATE, A_RAND, A_RANDN, A_KEEP, A_CHECK_KEPT, A_ASSIGN, A_FOREIGN_RANDINT, A_ENVSEED, \
    A_GETSTATE, A_IDENT, A_HOLD_RS, A_HELD_RANDINT, A_RUN_TO_DONE = range(18)
ALIAS_COLS = 8
ALIAS_KEPT_OBS_COL = 5          # A_CHECK_KEPT: CRC of the kept terminal observation -- survives reset() where frames are per-episode arrays (the reference;
#                                 this package with reference_dtypes=True), not where they are the engine's live buffers (the default uint8 frames)


def _bits(vec):
    return int(sum(int(b) << i for i, b in enumerate(np.asarray(vec).reshape(-1))))


def run_alias_script(env, ops, args, policy_seed=0):
    out = np.zeros((len(ops), ALIAS_COLS), dtype=np.int64)
    pol = np.random.RandomState(policy_seed)
    kept = foreign = held = last = None
    u8 = lambda a: crc(np.asarray(a).astype(np.uint8))  # noqa: E731
    # what the class returns: the Ray / OneHot classes a dict of four arrays (images, or one-hot states), the Flat class the bare frame
    ob = lambda o: o['observation'] if isinstance(o, dict) else o  # noqa: E731
    part = lambda o, k, dflt: o[k] if isinstance(o, dict) else dflt  # noqa: E731
    same = lambda o: isinstance(o, dict) and o['achieved_goal'] is o['observation']  # noqa: E731

    def one_step(a):
        o, r, d, info = env.step(a)
        ap = env.agent_pos
        return (o, info, d), [int(r), int(bool(d)), _bits(info['achieved_goal']), u8(ob(o)), int(env.step_num), ap.row * 256 + ap.col,
                              int(info['task_success'] is info['achieved_goal']) + 2 * int(same(o)), 0]
    for i, (op, arg) in enumerate(zip(ops, args)):
        op, arg = int(op), int(arg)
        if op == A_RESET:
            before = (env.desired_goal_vector, env.achieved_goal_vector)
            o = env.reset()
            st = env.np_random.get_state()
            row = [_bits(env.desired_goal_vector), u8(ob(o)), u8(part(o, 'desired_goal', env.desired_goal)), u8(part(o, 'init_observation', env.INIT_OBS)), int(st[2]),
                   crc(np.asarray(st[1], np.uint32)), int(env.ep_no),
                   int(env.desired_goal_vector is before[0]) + 2 * int(env.achieved_goal_vector is before[1]) + 4 * int(same(o))]
        elif op == A_STEP:
            last, row = one_step(arg)
        elif op == A_RUN_TO_DONE:                   # random actions (ids -6..5) until done; arg caps the number of steps; row = the last step's
            for _ in range(arg):
                last, row = one_step(int(pol.randint(-6, 6)))
                if last[2]:
                    break
        elif op == A_RANDINT:
            row = [int(env.np_random.randint(arg))]
        elif op == A_SHUFFLE:
            x = np.arange(arg, dtype=np.int64)
            env.np_random.shuffle(x)
            row = [crc(x)]
        elif op == A_RSEED:
            env.np_random.seed(arg)
            row = []
        elif op == A_RSETSTATE:
            env.np_random.set_state(np.random.RandomState(arg).get_state())
            row = []
        elif op == A_RAND:
            row = [int(env.np_random.rand() * 2 ** 53)]
        elif op == A_RANDN:
            row = [int(np.float64(env.np_random.randn()).view(np.int64))]
        elif op == A_KEEP:
            kept = last
            row = [_bits(kept[1]['achieved_goal']), _bits(kept[1]['desired_goal'])]
        elif op == A_CHECK_KEPT:
            ko, ki, _ = kept
            row = [_bits(ki['achieved_goal']), _bits(ki['desired_goal']), int(ki['achieved_goal'] is env.achieved_goal_vector),
                   int(ki['desired_goal'] is env.desired_goal_vector), int(ki['task_success'] is ki['achieved_goal']), u8(ob(ko))]
        elif op == A_ASSIGN:
            foreign = np.random.RandomState(arg)
            env.np_random = foreign
            row = [int(env.np_random is foreign)]
        elif op == A_FOREIGN_RANDINT:
            row = [int(foreign.randint(arg))]
        elif op == A_ENVSEED:
            old = env.np_random
            r = env.seed(arg)
            row = [int(env.np_random is old), int(r[0])]
        elif op == A_GETSTATE:
            st = env.np_random.get_state()
            row = [int(st[2]), crc(np.asarray(st[1], np.uint32)), int(st[3])]
        elif op == A_IDENT:
            row = [int(env.np_random is env.np_random), int(isinstance(env.np_random, np.random.RandomState))]
        elif op == A_HOLD_RS:
            held = env.np_random
            row = []
        elif op == A_HELD_RANDINT:
            row = [int(held.randint(arg))]
        else:
            raise ValueError('unknown op %d' % op)
        out[i, :len(row)] = row
    return out


def alias_script():
    """the ops of ray5_alias (the generator and the replaying tests build the same list; the fixture stores it too)"""
    S = []
    add = lambda op, arg=0: S.append((op, arg))  # noqa: E731
    add(A_IDENT)
    add(A_RESET)
    for a in (-1, -2, -3, -4, -5, -6, 4, -2, 5, -1):          # negative ids are list indices: -1 drop ... -6 up
        add(A_STEP, a)
    add(A_RUN_TO_DONE, 80); add(A_KEEP); add(A_RESET); add(A_CHECK_KEPT)     # the terminal info survives the reset
    add(A_RANDINT, 1000); add(A_RESET)                                       # a draw from np_random moves the next reset
    add(A_RUN_TO_DONE, 80); add(A_KEEP)
    add(A_SHUFFLE, 10); add(A_RAND); add(A_RESET); add(A_CHECK_KEPT)
    add(A_RSEED, 77); add(A_RESET)                                           # np_random.seed(77): in place
    add(A_STEP, 1); add(A_STEP, -3)
    add(A_RSETSTATE, 5); add(A_GETSTATE); add(A_RESET)                       # np_random.set_state(...)
    add(A_RANDN); add(A_GETSTATE); add(A_RUN_TO_DONE, 80); add(A_RESET); add(A_GETSTATE)   # a cached gaussian stays across the env's draws
    add(A_RANDN); add(A_GETSTATE)
    add(A_HOLD_RS); add(A_ENVSEED, 31); add(A_IDENT); add(A_RESET)           # env.seed(): a NEW generator, the old object is detached
    add(A_HELD_RANDINT, 1000); add(A_RESET)                                  # ... drawing from it does not move the env
    add(A_ASSIGN, 2024); add(A_RESET); add(A_FOREIGN_RANDINT, 10 ** 6)        # the caller's own RandomState becomes the env's generator
    add(A_RUN_TO_DONE, 80); add(A_KEEP); add(A_RESET); add(A_CHECK_KEPT); add(A_FOREIGN_RANDINT, 10 ** 6); add(A_GETSTATE)
    add(A_RANDINT, 50); add(A_RESET); add(A_IDENT)
    for a in (0, 1, 2, 3, -6, -5, -4, -3):
        add(A_STEP, a)
    return np.array([s[0] for s in S], np.int8), np.array([s[1] for s in S], np.int64)


def random_alias_script(rs, n):
    """a random op list of length n that run_alias_script can run (ops that need an earlier one -- a kept info, an assigned generator -- come after it)"""
    ops, args, have = [A_RESET], [0], dict(step=False, kept=False, foreign=False, held=False)
    menu = [A_RESET, A_STEP, A_STEP, A_STEP, A_RANDINT, A_SHUFFLE, A_RSEED, A_RSETSTATE, A_RAND, A_RANDN, A_KEEP, A_CHECK_KEPT, A_ASSIGN, A_FOREIGN_RANDINT,
            A_ENVSEED, A_GETSTATE, A_IDENT, A_HOLD_RS, A_HELD_RANDINT, A_RUN_TO_DONE]
    while len(ops) < n:
        op = int(rs.choice(menu))
        if (op == A_KEEP and not have['step']) or (op == A_CHECK_KEPT and not have['kept']) or (op == A_FOREIGN_RANDINT and not have['foreign']) or \
                (op == A_HELD_RANDINT and not have['held']):
            continue
        arg = {A_STEP: int(rs.randint(-6, 6)), A_RANDINT: int(rs.randint(1, 10 ** 6)), A_SHUFFLE: int(rs.randint(2, 40)), A_RSEED: int(rs.randint(1 << 31)),
               A_RSETSTATE: int(rs.randint(1 << 31)), A_ASSIGN: int(rs.randint(1 << 31)), A_FOREIGN_RANDINT: int(rs.randint(1, 10 ** 6)),
               A_ENVSEED: int(rs.randint(1 << 31)), A_HELD_RANDINT: int(rs.randint(1, 10 ** 6)), A_RUN_TO_DONE: int(rs.randint(1, 60))}.get(op, 0)
        have['step'] = have['step'] or op in (A_STEP, A_RUN_TO_DONE)
        have['kept'] = have['kept'] or op == A_KEEP
        have['foreign'] = have['foreign'] or op == A_ASSIGN
        have['held'] = have['held'] or op == A_HOLD_RS
        ops.append(op), args.append(arg)
    return np.array(ops, np.int8), np.array(args, np.int64)
