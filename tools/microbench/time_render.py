"""GPU box only: cw_render (mode 2, caller buffer) timed on its own, to separate the render kernel's rate from the
conditions inside a step sequence.   python time_render.py {rate|placement|interleave|preceding} [obs_mode]
  rate        back-to-back launches; in the pixel mode also whole steps
  placement   same kernel, six different buffers of one process (physical placement)
  interleave  own buffer vs engine buffer, after step kernels, after an idle gap
  preceding   which preceding kernel slows it (none / a torch kernel / cw_step)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from gym_craftingworld_amd import CraftingWorldVecEnv  # noqa: E402

N = 65536
what = sys.argv[1] if len(sys.argv) > 1 else 'rate'
mode = sys.argv[2] if len(sys.argv) > 2 else 'state'


def median_ms(fn, n=15, skip=3, pre=None):
    ts = []
    for i in range(n + skip):
        if pre:
            pre(i)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(i); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b))
    ts = sorted(ts[skip:])
    return ts[len(ts) // 2], ts[0]


acts = torch.randint(0, 6, (64, N), device='cuda', dtype=torch.uint8)
if what == 'rate':
    env = CraftingWorldVecEnv(N, obs_mode=mode, seed=0)
    env.reset()
    out = torch.empty((N, 84, 84, 3), dtype=torch.uint8, device='cuda')
    med, mn = median_ms(lambda i: env.render(out))
    print('cw_render(ext) median %.3f ms  %.0f GB/s   min %.3f' % (med, N * 21168 / med / 1e6, mn))
    if env.obs_mode == 'pixels':
        med, mn = median_ms(lambda i: env.step_async(acts[i % 64]), n=30, skip=10)
        print('cw_step(pixels) median %.3f ms min %.3f' % (med, mn))
elif what == 'placement':
    env = CraftingWorldVecEnv(N, obs_mode='state', seed=0)
    env.reset()
    bufs = [torch.empty((N, 84, 84, 3), dtype=torch.uint8, device='cuda') for _ in range(6)]
    for rep in range(2):
        for k, o in enumerate(bufs):
            print('buf %d addr %#x  %.3f ms' % (k, o.data_ptr(), median_ms(lambda i: env.render(o), n=9, skip=2)[0]))
elif what == 'interleave':
    env = CraftingWorldVecEnv(N, obs_mode='pixels_dirty', seed=0)
    env.reset()
    buf = torch.empty((N, 84, 84, 3), dtype=torch.uint8, device='cuda')
    step = lambda i: env.step_async(acts[i % 64])  # noqa: E731
    idle = lambda i: torch.cuda._sleep(600000)     # noqa: E731
    for name, out, pre in (('torch buf, back-to-back', buf, None), ('engine obs, back-to-back', env._obs, None),
                           ('engine init_obs, back-to-back', env._init_img, None), ('torch buf, after step kernels', buf, step),
                           ('engine obs, after step kernels', env._obs, step), ('torch buf, after 300us idle', buf, idle)):
        print('%-32s %.3f ms' % (name, median_ms(lambda i: env.render(out), n=10, skip=2, pre=pre)[0]))
elif what == 'preceding':
    buf = torch.empty((N, 84, 84, 3), dtype=torch.uint8, device='cuda')
    small = torch.zeros(1 << 20, device='cuda')
    for m, ar in (('state', False), ('state', True), ('pixels_dirty', True)):
        env = CraftingWorldVecEnv(N, obs_mode=m, seed=0, auto_reset=ar)
        env.reset()
        r = lambda i: env.render(buf)  # noqa: E731
        print('%-13s auto_reset=%d: none %.3f | torch add kernel %.3f | step %.3f' % (
            m, ar, median_ms(r, 10, 2)[0], median_ms(r, 10, 2, pre=lambda i: small.add_(1))[0],
            median_ms(r, 10, 2, pre=lambda i: env.step_async(acts[i % 64]))[0]))
        env.close()
else:
    raise SystemExit(__doc__)
