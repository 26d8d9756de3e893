"""Fast and slow launches of the one-launch step (they differ by ~15 us, tools/microbench/step_series.py): which waves make a launch slow?
Instrumented build (make -C gym_craftingworld_amd/csrc trace): every sweep wave stamps its start and end (100 MHz wall clock).
    python tools/microbench/render_waves2.py [pace]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ['CW_LIB_PATH'] = os.path.join(ROOT, 'gym_craftingworld_amd', 'libcraftingworld_trace.so')
import numpy as np  # noqa: E402
import torch  # noqa: E402
from gym_craftingworld_amd import CraftingWorldVecEnv, _lib  # noqa: E402

N = 65536
env = CraftingWorldVecEnv(N, size=(21, 21), max_steps=60000, obs_mode='pixels', seed=0)
lib = _lib.load()
env.reset()
acts = torch.randint(0, 6, (64, N), device='cuda', dtype=torch.uint8)
for t in range(300):
    env.step_async(acts[t % 64])
rows = []
for t in range(160):
    for k in range(4):                                   # keep the card busy: the traced launch is the last of 5 queued together
        env.step_async(acts[(t + k) % 64])
    env.step_async(acts[t % 64])
    torch.cuda.synchronize()
    buf = np.zeros((1024, 2), dtype=np.uint64)
    assert lib.cwk_trace_render_read(buf.ctypes.data_as(C.c_void_p)) == 0
    st, en = buf[:, 0].astype(np.int64), buf[:, 1].astype(np.int64)
    t0 = st.min()
    rows.append(((st - t0) / 100.0, (en - t0) / 100.0))
dur = np.array([e.max() for s, e in rows])
order = np.argsort(dur)
print('launch durations (first wave start -> last wave end), us: min %.1f p25 %.1f median %.1f p75 %.1f max %.1f' % (
    dur.min(), np.percentile(dur, 25), np.median(dur), np.percentile(dur, 75), dur.max()))
h, edges = np.histogram(dur, bins=12)
print('histogram:', ' '.join('%.0f:%d' % (edges[i], h[i]) for i in range(len(h))))
blk = np.arange(1024) // 4


def describe(tag, idx):
    s, e = rows[idx]
    late = np.argsort(e)[-24:]
    print('%s launch %3d: %.1f us | starts: median %.1f p99 %.1f max %.1f | ends: p1 %.1f p10 %.1f median %.1f p90 %.1f p99 %.1f max %.1f | busy: median %.1f max %.1f' % (
        tag, idx, dur[idx], np.median(s), np.percentile(s, 99), s.max(), np.percentile(e, 1), np.percentile(e, 10), np.median(e), np.percentile(e, 90),
        np.percentile(e, 99), e.max(), np.median(e - s), (e - s).max()))
    print('      last 24 waves: XCD (workgroup %% 8) counts %s; their starts: median %.1f max %.1f; distinct workgroups %d' % (
        np.bincount(blk[late] % 8, minlength=8).tolist(), np.median(s[late]), s[late].max(), len(set(blk[late].tolist()))))
    print('      mean end by XCD: ' + ' '.join('%.1f' % e[(blk % 8) == k].mean() for k in range(8)))


for i in order[:4]:
    describe('FAST', int(i))
for i in order[-6:]:
    describe('SLOW', int(i))
fast = [rows[i] for i in order[:40]]
slow = [rows[i] for i in order[-40:]]
print('mean over the 40 fastest / 40 slowest launches: median end %.1f / %.1f, p99 end %.1f / %.1f, max start %.1f / %.1f, median busy %.1f / %.1f' % (
    np.mean([np.median(e) for s, e in fast]), np.mean([np.median(e) for s, e in slow]),
    np.mean([np.percentile(e, 99) for s, e in fast]), np.mean([np.percentile(e, 99) for s, e in slow]),
    np.mean([s.max() for s, e in fast]), np.mean([s.max() for s, e in slow]),
    np.mean([np.median(e - s) for s, e in fast]), np.mean([np.median(e - s) for s, e in slow])))
env.close()
