"""Load the golden fixtures captured from the reference (tools/gen_golden.py) and replay them."""
import glob
import json
import os
import zlib

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def fixture_names():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN, '*.npz')))


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
