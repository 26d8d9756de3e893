"""Per-step wall time (HIP events) of cw_step: median, and the synchronized time-out steps.
    python tools/microbench/time_steps.py [obs_mode] [n_envs] [sync|desync] [graph]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gym_craftingworld_amd import CraftingWorldVecEnv

mode = sys.argv[1] if len(sys.argv) > 1 else 'state'
N = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
env = CraftingWorldVecEnv(N, size=(21, 21), max_steps=300, obs_mode=mode, seed=0)
env.reset()
desync = len(sys.argv) > 3 and sys.argv[3] == 'desync'
if desync:      # steady state of a long run: episode phases spread out, ~N/300 envs finish on every step
    import numpy as np
    env.set_state(step_num=(np.arange(N) * 7 % 300).astype(np.int32))
use_graph = len(sys.argv) > 4 and sys.argv[4] == 'graph'
T = 640
acts = torch.randint(0, 6, (T, N), device='cuda', dtype=torch.uint8)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(T + 1)]
for t in range(20):
    env.step(acts[t])
torch.cuda.synchronize()
if use_graph:     # GPU-side time without the host's launch path: 16 steps per graph, events around each replay
    G = 16
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for t in range(G):
            env.step_async(acts[t])
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for t in range(G):
            env.step_async(acts[t])
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    R = T // G
    ev[0].record()
    for r in range(R):
        graph.replay()
        ev[r + 1].record()
    torch.cuda.synchronize()
    ms = torch.tensor([ev[r].elapsed_time(ev[r + 1]) / G for r in range(R)])
    srt = ms.sort().values
    print('%s N=%d %s fused=%s graph of %d steps: per step median %.2f us, p90 %.2f us, mean %.2f us' % (
        mode, N, 'desync' if desync else 'sync', os.environ.get('CW_TUNE_LOOKAHEAD', '1'), G,
        1e3 * srt[R // 2], 1e3 * srt[int(R * 0.9)], 1e3 * ms.mean()))
    env.close()
    sys.exit(0)
ev[0].record()
for t in range(T):
    env.step(acts[t])
    ev[t + 1].record()
torch.cuda.synchronize()
ms = torch.tensor([ev[t].elapsed_time(ev[t + 1]) for t in range(T)])
srt = ms.sort().values
top = ms.topk(4)
print('%s N=%d %s fused=%s: median %.1f us, p90 %.1f us, mean %.1f us, slowest steps %s' % (
    mode, N, 'desync' if desync else 'sync', os.environ.get('CW_TUNE_LOOKAHEAD', '1'), 1e3 * srt[T // 2], 1e3 * srt[int(T * 0.9)], 1e3 * ms.mean(),
    ', '.join('t=%d: %.0f us' % (20 + i, 1e3 * v) for v, i in zip(top.values.tolist(), top.indices.tolist()))))
env.close()
