"""One-off fuzz of the full-frame step against the dirty-cell engine (GPU box): random grid sizes (4x4 ... 100x100, i.e. 2 to 9 frames per piece), batch
sizes, rasters, episode lengths, chunk sizes, clocked / unclocked sweeps, look-ahead records on / off.  python tools/microbench/fuzz_pieces.py [n_cases]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, '.')
from gym_craftingworld_amd import CraftingWorldVecEnv
rng = np.random.RandomState(int(os.environ.get('FUZZ_SEED', '7')))
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
bad = 0
for case in range(n_cases):
    raster = 'alt' if rng.rand() < 0.4 else 'ray'
    S = int(rng.choice([4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 16, 17, 21, 23, 29, 32, 40, 47, 64, 70, 100]))
    fb = 27 * S * (S + 1) if raster == 'alt' else 48 * S * S
    N = int(min(rng.choice([1, 2, 3, 7, 64, 65, 333, 1000, 1024, 1025, 4099, 9000, 20000]), (1 << 30) // fb))
    max_steps = int(rng.choice([3, 5, 7, 11, 19]))
    os.environ['CW_TUNE_PERIOD_NS'] = str(int(rng.choice([0, 300, 600, 2000])))
    rng.choice([0, 1, 4])                                   # (a draw of the first runs' seeds: the pace inside a job, a knob that is gone)
    os.environ['CW_TUNE_LOOKAHEAD'] = str(int(rng.rand() < 0.8))
    if rng.rand() < 0.4:
        os.environ['CW_TUNE_RENDER_CHUNK_ROUNDS'] = '1'
    else:
        os.environ.pop('CW_TUNE_RENDER_CHUNK_ROUNDS', None)
    kw = dict(size=(S, S), max_steps=max_steps, seed=int(rng.randint(1 << 30)), raster=raster)
    full = CraftingWorldVecEnv(N, obs_mode='pixels', **kw)
    dirty = CraftingWorldVecEnv(N, obs_mode='pixels_dirty', **kw)
    full.reset(); dirty.reset()
    gen = torch.Generator(device='cuda').manual_seed(case)
    ok = True
    for t in range(3 * max_steps + 20):
        a = torch.randint(0, 6, (N,), device='cuda', dtype=torch.uint8, generator=gen)
        if t == max_steps + 1:
            for e in (full, dirty):
                e.set_state(step_num=(np.arange(N) % max_steps).astype(np.int32))
        if t == max_steps + 2:
            full._obs.fill_(9)
        of, rf, df, _ = full.step(a); od, rd, dd, _ = dirty.step(a)
        for k in ('observation', 'desired_goal', 'init_observation'):
            if not torch.equal(of[k], od[k]): ok = False
        if not (torch.equal(rf, rd) and torch.equal(df, dd)): ok = False
    if not torch.equal(full.render(), dirty.render()): ok = False
    kf, pf = full.get_rng_states(); kd, pd = dirty.get_rng_states()
    if not (np.array_equal(kf, kd) and np.array_equal(pf, pd)): ok = False
    bad += not ok
    print('%3d %s %-4s S=%-3d N=%-5d max_steps=%-2d chunked=%s period=%s lookahead=%s' % (case, 'ok ' if ok else 'BAD', raster, S, N, max_steps,
          'CW_TUNE_RENDER_CHUNK_ROUNDS' in os.environ, os.environ['CW_TUNE_PERIOD_NS'], os.environ['CW_TUNE_LOOKAHEAD']), flush=True)
    full.close(); dirty.close()
print('cases', n_cases, 'bad', bad)
sys.exit(1 if bad else 0)
