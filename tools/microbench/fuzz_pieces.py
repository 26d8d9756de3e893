"""One-off fuzz of the full-frame painters against the dirty-cell engine (GPU box): random grid sizes, batch sizes, rasters, episode lengths,
chunk sizes; the piece sweep forced where the frames allow it.  python tools/microbench/fuzz_pieces.py [n_cases]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, '.')
os.environ['CW_EXPERIMENT_BUILD'] = '1'
from gym_craftingworld_amd import CraftingWorldVecEnv
rng = np.random.RandomState(int(os.environ.get('FUZZ_SEED', '7')))
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
bad = 0
for case in range(n_cases):
    raster = 'alt' if rng.rand() < 0.4 else 'ray'
    S = int(rng.choice([10, 11, 12, 13, 16, 17, 21, 23, 29, 32, 40, 47, 64, 70, 100]))
    fb = 27 * S * (S + 1) if raster == 'alt' else 48 * S * S
    N = int(min(rng.choice([1, 2, 3, 7, 64, 65, 333, 1000, 1024, 1025, 4099, 9000]), (1 << 30) // fb))
    max_steps = int(rng.choice([3, 5, 7, 11]))
    os.environ['CW_TUNE_PIECE_PACE'] = str(int(rng.choice([0, 1, 2, 4])))
    if rng.rand() < 0.4:
        os.environ['CW_TUNE_RENDER_CHUNK_ROUNDS'] = str(int(rng.choice([1, 2, 5])))
    else:
        os.environ.pop('CW_TUNE_RENDER_CHUNK_ROUNDS', None)
    arrangement = rng.choice(['one launch', 'two streams', 'one stream'])
    for v in ('CW_TUNE_FUSED_RENDER', 'CW_TUNE_OVERLAP'):
        os.environ.pop(v, None)
    if arrangement == 'two streams': os.environ['CW_TUNE_FUSED_RENDER'] = '0'
    if arrangement == 'one stream': os.environ['CW_TUNE_OVERLAP'] = '0'
    kw = dict(size=(S, S), max_steps=max_steps, seed=int(rng.randint(1 << 30)), raster=raster)
    full = CraftingWorldVecEnv(N, obs_mode='pixels', **kw)
    dirty = CraftingWorldVecEnv(N, obs_mode='pixels_dirty', **kw)
    name = full.render_kernel_name()
    full.reset(); dirty.reset()
    gen = torch.Generator(device='cuda').manual_seed(case)
    ok = True
    for t in range(3 * max_steps + 4):
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
    bad += not ok
    print('%3d %s %-4s S=%-3d N=%-5d max_steps=%-2d %-11s %-30s chunk=%s pace=%s' % (case, 'ok ' if ok else 'BAD', raster, S, N, max_steps, arrangement, name,
          os.environ.get('CW_TUNE_RENDER_CHUNK_ROUNDS', '-'), os.environ['CW_TUNE_PIECE_PACE']), flush=True)
    full.close(); dirty.close()
print('cases', n_cases, 'bad', bad)
sys.exit(1 if bad else 0)
