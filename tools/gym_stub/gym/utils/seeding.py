"""gym<=0.21 `seeding.np_random` restated from its published algorithm (UNPINNED: gym is not
installed, so this mapping seed -> MT19937 key cannot be verified offline; golden fixtures
therefore inject raw RandomState states and never depend on this function)."""
import hashlib
import os
import struct

import numpy as np


def _bigint_from_bytes(b):
    sizeof_int = 4
    padding = sizeof_int - len(b) % sizeof_int
    b += b'\0' * padding
    int_count = len(b) // sizeof_int
    unpacked = struct.unpack('{}I'.format(int_count), b)
    return sum(2 ** (sizeof_int * 8 * i) * val for i, val in enumerate(unpacked))


def create_seed(a=None, max_bytes=8):
    if a is None:
        return _bigint_from_bytes(os.urandom(max_bytes))
    return int(a) % 2 ** (8 * max_bytes)


def hash_seed(seed=None, max_bytes=8):
    if seed is None:
        seed = create_seed(max_bytes=max_bytes)
    return _bigint_from_bytes(hashlib.sha512(str(seed).encode('utf8')).digest()[:max_bytes])


def _int_list_from_bigint(bigint):
    if bigint == 0:
        return [0]
    ints = []
    while bigint > 0:
        bigint, mod = divmod(bigint, 2 ** 32)
        ints.append(mod)
    return ints


def np_random(seed=None):
    seed = create_seed(seed)
    rng = np.random.RandomState()
    rng.seed(_int_list_from_bigint(hash_seed(seed)))
    return rng, seed
