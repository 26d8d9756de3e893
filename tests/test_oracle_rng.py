"""The oracle's MT19937 / randint / shuffle restatement vs the installed numpy RandomState
(numpy is the third-party dependency that owns this arithmetic)."""
import numpy as np
import pytest

from oracle import OracleEnv


def _pair(seed):
    rs = np.random.RandomState(seed)
    st = rs.get_state()
    env = OracleEnv(rng_state=(st[1], st[2]), size=(5, 5))
    return rs, env


@pytest.mark.parametrize('seed', [0, 1, 12345, 2**32 - 1])
def test_raw_stream_and_seed(seed):
    rs, env = _pair(seed)
    e2 = OracleEnv(size=(5, 5))
    e2.seed_int(seed)                       # init_genrand == RandomState(int)
    ref = rs.randint(0, 2**32, size=2000, dtype=np.uint32)  # full-range uint32 = raw genrand
    got = np.array([env.rng_u32() for _ in range(2000)], dtype=np.uint32)
    got2 = np.array([e2.rng_u32() for _ in range(2000)], dtype=np.uint32)
    assert np.array_equal(ref, got) and np.array_equal(ref, got2)


@pytest.mark.parametrize('seed', range(5))
def test_randint_and_shuffle_interleaved(seed):
    rs, env = _pair(seed)
    for n in [1, 2, 3, 5, 9, 16, 17, 441, 1024, 1, 432, 1000003]:
        assert int(rs.randint(n)) == env.rng_randint(n), n
        m = (n % 50) + 1
        x = np.arange(m)
        rs.shuffle(x)
        assert np.array_equal(x, env.rng_shuffle(m)), m
    key, pos = env.get_rng()
    st = rs.get_state()
    assert pos == st[2] and np.array_equal(key, st[1])
