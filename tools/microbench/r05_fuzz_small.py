"""GPU box: fuzz of round 5's painters against the dirty-cell engine -- SMALL frames in batches large enough for several workgroups per CU (the gather painter
and the piece sweep, 1 / 2 / 4 / 8 workgroups per CU, gather up to 7x7 or 9x9), the step kernel's cooperative painter with and without terminal frames,
look-ahead records on / off, clocked / unclocked / chunked sweeps.   FUZZ_SEED=n python tools/microbench/r05_fuzz_small.py [n_cases]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, '.')
from gym_craftingworld_amd import CraftingWorldVecEnv
rng = np.random.RandomState(int(os.environ.get('FUZZ_SEED', '11')))
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
bad = 0
for case in range(n_cases):
    raster = 'alt' if rng.rand() < 0.3 else 'ray'
    S = int(rng.choice([4, 5, 6, 7, 8, 9, 10, 11, 12, 21]))
    fb = 27 * S * (S + 1) if raster == 'alt' else 48 * S * S
    N = int(min(rng.choice([1, 63, 64, 65, 255, 257, 1000, 4097, 9001, 20000, 40001, 70000]), (3 << 29) // fb))
    max_steps = int(rng.choice([2, 3, 5, 7, 11]))
    env = {'CW_TUNE_PERIOD_NS': str(int(rng.choice([0, 300, 600, 2000]))), 'CW_TUNE_LOOKAHEAD': str(int(rng.rand() < 0.8)),
           'CW_TUNE_SMALL_BLOCKS': str(int(rng.choice([1, 2, 4, 8]))), 'CW_TUNE_GATHER': str(int(rng.rand() < 0.7)),
           'CW_TUNE_GATHER_MAX_SIZE': str(int(rng.choice([7, 9]))), 'CW_TUNE_STEP_ENVS_PER_WAVE': str(int(rng.choice([8, 16, 32, 64])))}
    if rng.rand() < 0.3:
        env['CW_TUNE_RENDER_CHUNK_ROUNDS'] = '1'
    for k in list(os.environ):
        if k.startswith('CW_TUNE_'):
            del os.environ[k]
    os.environ.update(env)
    keep = bool(rng.rand() < 0.4)
    kw = dict(size=(S, S), max_steps=max_steps, seed=int(rng.randint(1 << 30)), raster=raster, keep_terminal_obs=keep)
    full = CraftingWorldVecEnv(N, obs_mode='pixels', **kw)
    dirty = CraftingWorldVecEnv(N, obs_mode='pixels_dirty', **kw)
    full.reset(); dirty.reset()
    gen = torch.Generator(device='cuda').manual_seed(case)
    ok = True
    for t in range(3 * max_steps + 12):
        a = torch.randint(0, 6, (N,), device='cuda', dtype=torch.uint8, generator=gen)
        if t == max_steps + 1:
            for e in (full, dirty):
                e.set_state(step_num=(np.arange(N) % max_steps).astype(np.int32))
        if t == max_steps + 2:
            full._obs.fill_(9)
        of, rf, df, inf = full.step(a); od, rd, dd, ind = dirty.step(a)
        for k in ('observation', 'desired_goal', 'init_observation'):
            if not torch.equal(of[k], od[k]): ok = False
        if not (torch.equal(rf, rd) and torch.equal(df, dd)): ok = False
        if not (torch.equal(inf['episode']['r'][df], ind['episode']['r'][dd]) and torch.equal(inf['episode']['l'][df], ind['episode']['l'][dd])): ok = False
        if keep and not torch.equal(inf['terminal_observation'][df], ind['terminal_observation'][dd]): ok = False
    if not torch.equal(full.render(), dirty.render()): ok = False
    kf, pf = full.get_rng_states(); kd, pd = dirty.get_rng_states()
    if not (np.array_equal(kf, kd) and np.array_equal(pf, pd) and torch.equal(full.counters, dirty.counters)): ok = False
    bad += not ok
    print('%3d %s %-4s S=%-3d N=%-5d max_steps=%-2d terminal=%d %s %s' % (case, 'ok ' if ok else 'BAD', raster, S, N, max_steps, keep, full.render_kernel_name(),
          ' '.join('%s=%s' % (k[8:], v) for k, v in sorted(env.items()))), flush=True)
    full.close(); dirty.close()
print('cases', n_cases, 'bad', bad)
sys.exit(1 if bad else 0)
