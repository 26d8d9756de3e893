"""CPU tier, SURVEY 5 ("race detection / sanitizers": ASAN build of the CPU oracle + -fsanitize=address host side): what can run under a sanitizer
without a GPU does, each in a subprocess with libasan preloaded --
  * the C oracle (make -C oracle asan: -fsanitize=address,undefined) replays reference fixtures of every kind (CW_ORACLE_SO picks the build);
  * the engine's HIP-free host logic (csrc/cw_host.cpp by g++ -fsanitize=address,undefined -fno-sanitize-recover: libcw_host_asan.so; CW_HOST_LIB
    picks it) runs the hypothesis properties of the MT19937 state conversion and rewind, the guard's synthetic traces, the dense views and the
    checkpoint-size arithmetic.
GPU AddressSanitizer is not available on the pool: the kernels' memory safety is argued by the parity tests reading every byte they write."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _libasan():
    p = subprocess.run(['gcc', '-print-file-name=libasan.so'], capture_output=True, text=True).stdout.strip()
    if not p or not os.path.isabs(p) or not os.path.exists(p):
        pytest.skip('no libasan beside this gcc')
    return os.path.realpath(p)


def _pytest_under_asan(extra_env, args):
    env = dict(os.environ, LD_PRELOAD=_libasan(), ASAN_OPTIONS='detect_leaks=0:abort_on_error=1', UBSAN_OPTIONS='halt_on_error=1:print_stacktrace=1',
               PYTHONDONTWRITEBYTECODE='1', **extra_env)
    p = subprocess.run([sys.executable, '-m', 'pytest', '-x', '-q', '-p', 'no:cacheprovider'] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    tail = (p.stdout + p.stderr)[-3000:]
    assert p.returncode == 0, tail
    assert 'ERROR: AddressSanitizer' not in tail and 'runtime error:' not in tail, tail
    return p.stdout


def test_oracle_replays_fixtures_under_asan_and_ubsan():
    subprocess.check_call(['make', '-C', os.path.join(ROOT, 'oracle'), 'asan'], stdout=subprocess.DEVNULL)
    so = os.path.join(ROOT, 'oracle', 'libcw_oracle_asan.so')
    out = _pytest_under_asan({'CW_ORACLE_SO': so}, ['tests/test_oracle_golden.py', 'tests/test_oracle_rng.py', '-k',
                                                    'ray5_random or ray8_subset_sel or ray6_fixedinit or alt4_double or onehot5_random or flat8_random or rng or shuffle or randint'])
    assert ' passed' in out and 'failed' not in out, out
    # (the library the subprocess loaded IS the instrumented one: it names the sanitizer runtime among its dependencies)
    assert 'libasan' in subprocess.run(['ldd', so], capture_output=True, text=True).stdout


def test_host_logic_under_asan_and_ubsan():
    subprocess.check_call(['make', '-C', os.path.join(ROOT, 'gym_craftingworld_amd', 'csrc'), 'host_asan'], stdout=subprocess.DEVNULL)
    so = os.path.join(ROOT, 'gym_craftingworld_amd', 'libcw_host_asan.so')
    out = _pytest_under_asan({'CW_HOST_LIB': so}, ['tests/test_guard_logic.py', 'tests/test_host_logic.py', '-k',
                                                   'guard or steady or late or alternating or saturated or probes or delay or disturbance or periods or dense or refill_period or '
                                                   'checkpoint_section or mt_conversion_property or mt_rewind_property'])
    assert ' passed' in out and 'failed' not in out, out
    n = int(out.strip().splitlines()[-1].split(' passed')[0].split()[-1])
    assert n >= 15, out
