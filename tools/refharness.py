"""Build-container-only harness: imports the read-only reference (/root/reference) under the
stand-in `gym` package in tools/gym_stub and exposes helpers shared by gen_golden.py and
diff_vs_reference.py.  Never imported by the product, tests/ or bench.py (the reference does
not exist on the GPU box)."""
import os
import sys
import zlib
from collections import deque

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE = '/root/reference'


def import_reference():
    os.environ.setdefault('MPLBACKEND', 'Agg')
    sys.dont_write_bytecode = True
    for p in (os.path.join(HERE, 'gym_stub'), REFERENCE):
        if p not in sys.path:
            sys.path.insert(0, p)
    import gym  # noqa: F401  (the stub)
    import gym_craftingworld  # noqa: F401  (registers the ids)
    from gym_craftingworld.envs import (CraftingWorldEnvAltObs, CraftingWorldEnvFlat, CraftingWorldEnvOneHot,
                                        CraftingWorldEnvRay)
    return dict(ray=CraftingWorldEnvRay, flat=CraftingWorldEnvFlat, onehot=CraftingWorldEnvOneHot,
                altobs=CraftingWorldEnvAltObs)


def make_ref_env(cls, rng, **kwargs):
    """Construct a reference env whose np_random is `rng` from the first draw on (the ctor's
    generate_fixed_states draws from self.np_random, ray.py:116-118, so the RNG has to be in
    place before __init__ runs: patch the stub's seeding hook for the duration of the ctor)."""
    from gym.utils import seeding
    orig = seeding.np_random
    seeding.np_random = lambda seed=None: (rng, 0)
    try:
        env = cls(**kwargs)
    finally:
        seeding.np_random = orig
    assert env.np_random is rng
    return env


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


def codes_from_onehot(oh):
    """(H,W,12) reference one-hot -> (codes uint8 (H,W), agent (r,c), hold 0..3)."""
    oh = np.asarray(oh)
    codes = (oh[:, :, :8] * np.arange(1, 9)).sum(axis=2).astype(np.uint8)
    assert (oh[:, :, :8].sum(axis=2) <= 1).all(), 'more than one object in a cell'
    ar, ac = np.where(oh[:, :, 8] == 1)
    assert len(ar) == 1
    h = oh[ar[0], ac[0], 9:]
    hold = int(np.argmax(h)) + 1 if h.any() else 0
    return codes, (int(ar[0]), int(ac[0])), hold


def bits(vec):
    v = np.asarray(vec).reshape(-1)
    return int(sum(int(b) << i for i, b in enumerate(v)))


# ---------------------------------------------------------------- scripted skill-seeking policy
STICKS, AXE, HAMMER, ROCK, TREE, BREAD, HOUSE, WHEAT = range(1, 9)
MOVES = [(-1, 0), (0, 1), (1, 0), (0, -1)]  # action ids 0..3 = up,right,down,left (ray.py:130-131)


def _bfs_first_action(codes, agent, hold, target, avoid=()):
    """First action of a shortest path agent->target cell; cells holding a code in `avoid`, rocks
    without hammer and trees without axe are walls (target itself exempt from `avoid`)."""
    n = codes.shape[0]
    blocked = lambda r, c: ((codes[r, c] == ROCK and hold != 3) or (codes[r, c] == TREE and hold != 2))
    prev = {agent: None}
    q = deque([agent])
    while q:
        cur = q.popleft()
        if cur == target:
            break
        for a, (dr, dc) in enumerate(MOVES):
            nr, nc = cur[0] + dr, cur[1] + dc
            if not (0 <= nr < n and 0 <= nc < n) or (nr, nc) in prev:
                continue
            if blocked(nr, nc) or ((nr, nc) != target and codes[nr, nc] in avoid):
                continue
            prev[(nr, nc)] = (cur, a)
            q.append((nr, nc))
    if target not in prev or target == agent:
        return None
    cur = target
    while prev[cur][0] != agent:
        cur = prev[cur][0]
    return prev[cur][1]


def scripted_action(env, rng):
    """Goal-directed policy on the reference env's own state: works towards the first desired
    task not yet achieved; falls back to a random action.  Only used to make the golden
    trajectories visit skill completions / successes on large grids."""
    codes, agent, hold = codes_from_onehot(env.obs_one_hot)
    want = env.desired_goal_vector[0]
    have = env.achieved_goal_vector[0]
    find = lambda code: [tuple(p) for p in np.argwhere(codes == code)]

    def goto(target, avoid=()):
        a = _bfs_first_action(codes, agent, hold, target, avoid)
        return a if a is not None else int(rng.randint(6))

    def fetch(tool):  # tool: 1 sticks, 2 axe, 3 hammer
        if hold == tool:
            return None
        if hold != 0:
            return 5 if codes[agent] == 0 else int(rng.randint(4))   # drop (needs an empty cell)
        locs = find(tool)
        if not locs:
            return int(rng.randint(6))
        return 4 if agent == locs[0] else goto(locs[0], avoid=(BREAD, WHEAT, STICKS) if tool != 1 else (BREAD,))

    order = [0, 3, 4, 2, 1, 6, 7, 8, 5]  # MakeBread, ChopTree, ChopRock, BuildHouse, EatBread, moves, GoToHouse
    for t in order:
        if t >= len(want) or not want[t] or have[t]:
            continue
        if t == 0:
            a = fetch(2)
            return a if a is not None else (goto(find(WHEAT)[0]) if find(WHEAT) else int(rng.randint(6)))
        if t == 3:
            a = fetch(2)
            return a if a is not None else (goto(find(TREE)[0]) if find(TREE) else int(rng.randint(6)))
        if t == 4:
            a = fetch(3)
            return a if a is not None else (goto(find(ROCK)[0]) if find(ROCK) else int(rng.randint(6)))
        if t == 2:
            a = fetch(3)
            return a if a is not None else (goto(find(STICKS)[0]) if find(STICKS) else int(rng.randint(6)))
        if t == 1:
            return goto(find(BREAD)[0]) if find(BREAD) else int(rng.randint(6))
        if t in (6, 7, 8):
            a = fetch({6: 2, 7: 3, 8: 1}[t])
            return a if a is not None else int(rng.randint(4))
        if t == 5:
            return goto(find(HOUSE)[0]) if find(HOUSE) else int(rng.randint(6))
    return int(rng.randint(6))
