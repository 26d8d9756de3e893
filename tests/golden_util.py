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
    return meta, kw, d


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF
