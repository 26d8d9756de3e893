"""Where a reset's latency goes: 100 MHz wall-clock stamps inside reset_env_wave (trace build, `make -C
gym_craftingworld_amd/csrc trace`).  Phases: 0->1 MT state to LDS, 1->2 task draw, 2->3 placement shuffle,
3->4 imagine_obs, 4->5 MT write-back.  Prints medians for (a) the all-env reset (chip full of reset waves) and
(b) sparse auto-resets between steps.
    python tools/microbench/reset_phases.py [n_envs]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ['CW_LIB_PATH'] = os.path.join(ROOT, 'gym_craftingworld_amd', 'libcraftingworld_trace.so')
os.environ.setdefault('CW_TUNE_LOOKAHEAD', '0')      # (the slow path: resets inline in the step kernel, where the stamps are)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from gym_craftingworld_amd import CraftingWorldVecEnv, _lib  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
mode = sys.argv[2] if len(sys.argv) > 2 else 'state'
env = CraftingWorldVecEnv(N, size=(21, 21), max_steps=300, obs_mode=mode, seed=0)
lib = _lib.load()


def snap():
    torch.cuda.synchronize()
    buf = np.zeros((1024, 8), dtype=np.uint64)
    assert lib.cwk_trace_read(buf.ctypes.data_as(C.c_void_p)) == 0
    return buf


def report(tag, rows):
    d = np.diff(rows[:, :6].astype(np.int64), axis=1) * 10.0 / 1e3     # us
    tot = (rows[:, 5].astype(np.int64) - rows[:, 0].astype(np.int64)) * 10.0 / 1e3
    names = ['load', 'task', 'shuffle', 'imagine', 'store']
    cyc = (rows[:, 7].astype(np.int64) - rows[:, 6].astype(np.int64))
    print('%s (%d resets): total median %.2f us, p90 %.2f | ' % (tag, len(rows), np.median(tot), np.percentile(tot, 90)) +
          ', '.join('%s %.2f' % (n, np.median(d[:, i])) for i, n in enumerate(names)) +
          ' | s_memtime ticks per us %.0f' % np.median(cyc / np.maximum(tot, 1e-3)))


env.reset()
a = snap()
report('all-env reset, N=%d' % N, a)
phase = (np.arange(N) * 7 % 300).astype(np.int32)
env.set_state(step_num=phase)
acts = torch.randint(0, 6, (8, N), device='cuda', dtype=torch.uint8)
env.step(acts[0])
b0 = snap()
for t in range(1, 8):
    env.step(acts[t])
b = snap()
ch = (b[:, 0] != b0[:, 0])
report('sparse auto-reset (%s mode, ~%d done per step)' % (mode, N // 300), b[ch])
env.close()
