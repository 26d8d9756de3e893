"""seed(int) -> MT19937 state, host side (replaces gym.utils.seeding.np_random at ray.py:146).

UNPINNED: the reference leaves gym unpinned (requirements.txt:1) and gym is not installable here,
so the mapping below is the published gym<=0.21 algorithm restated (sha512 of str(seed), first 8
bytes, little-endian uint32 limbs -> numpy init_by_array) and cannot be checked against gym
offline.  Bit-exact parity is therefore defined at the MT19937-state level (set_rng_states);
this function is a convenience on top of it.
"""
import hashlib
import os
import struct

import numpy as np


def _bigint_from_bytes(b):
    b = b + b'\0' * (4 - len(b) % 4)
    limbs = struct.unpack('{}I'.format(len(b) // 4), b)
    return sum(v << (32 * i) for i, v in enumerate(limbs))


def create_seed(a=None, max_bytes=8):
    if a is None:
        return _bigint_from_bytes(os.urandom(max_bytes))
    if not (isinstance(a, (int, np.integer)) and a >= 0):
        raise ValueError('Seed must be a non-negative integer or omitted, not {}'.format(a))
    return int(a) % 2 ** (8 * max_bytes)


def hash_seed(seed, max_bytes=8):
    return _bigint_from_bytes(hashlib.sha512(str(seed).encode('utf8')).digest()[:max_bytes])


def mt_state_from_seed(seed):
    """-> (key uint32[624], pos) of numpy RandomState seeded the gym<=0.21 way."""
    h = hash_seed(seed)
    limbs = []
    while h > 0:
        h, mod = divmod(h, 2 ** 32)
        limbs.append(mod)
    rs = np.random.RandomState()
    rs.seed(limbs or [0])
    st = rs.get_state()
    return st[1].astype(np.uint32), int(st[2])


def np_random(seed=None):
    """gym<=0.21 seeding.np_random(seed) -> (RandomState, seed)."""
    seed = create_seed(seed)
    key, pos = mt_state_from_seed(seed)
    rs = np.random.RandomState()
    rs.set_state(('MT19937', key, pos, 0, 0.0))
    return rs, seed
