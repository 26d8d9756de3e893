"""When do the render kernel's 1024 waves start and finish within one launch?  (instrumented build: make trace)
    python tools/microbench/render_waves.py"""
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
env = CraftingWorldVecEnv(N, size=(21, 21), max_steps=300, obs_mode='pixels', seed=0)
lib = _lib.load()
env.reset()
env.set_state(step_num=(np.arange(N) * 7 % 300).astype(np.int32))
acts = torch.randint(0, 6, (40, N), device='cuda', dtype=torch.uint8)
all_busy = []
for t in range(40):
    env.step(acts[t])
    if t >= 30:
        torch.cuda.synchronize()
        buf = np.zeros((1024, 2), dtype=np.uint64)
        assert lib.cwk_trace_render_read(buf.ctypes.data_as(C.c_void_p)) == 0
        st, en = buf[:, 0].astype(np.int64), buf[:, 1].astype(np.int64)
        t0 = st.min()
        e = (en - t0) / 100.0
        s_ = (st - t0) / 100.0
        all_busy.append(e - s_)
        print('launch %2d: waves start %.1f..%.1f us; finish min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f us; mean busy %.1f us' % (
            t, s_.min(), s_.max(), e.min(), np.percentile(e, 10), np.median(e), np.percentile(e, 90), e.max(), (e - s_).mean()))
busy = np.stack(all_busy)            # [launch, wave]
print('correlation of per-wave busy time between consecutive launches: %.2f' % np.mean([np.corrcoef(busy[i], busy[i + 1])[0, 1] for i in range(len(busy) - 1)]))
m = busy.mean(axis=0)
blk = np.arange(1024) // 4
print('mean busy by block %% 8 (XCD if workgroups go round-robin): ' + ' '.join('%.1f' % m[(blk % 8) == k].mean() for k in range(8)))
print('mean busy by wave-in-block: ' + ' '.join('%.1f' % m[np.arange(1024) % 4 == k].mean() for k in range(4)))
print('mean busy by block // 8 %% 4: ' + ' '.join('%.1f' % m[((blk // 8) % 4) == k].mean() for k in range(4)))
srt = np.sort(m)
print('per-wave mean busy over %d launches: min %.1f p10 %.1f median %.1f p90 %.1f max %.1f us' % (len(busy), srt[0], srt[102], srt[512], srt[921], srt[-1]))
env.close()
