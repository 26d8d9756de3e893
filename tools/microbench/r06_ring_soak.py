"""GPU box: the ring of look-ahead records under heavy use against an engine that keeps none -- the walker of r06_short_episodes.py (480 of 65 536 envs finishing
per step, many of them several times per refill period, the period adapting) for 6 000 steps, and episodes of at most 3 steps (every ring runs empty all the time):
state, counters and every env's RNG stream must be equal at the end.   python tools/microbench/r06_ring_soak.py"""
import os, sys
sys.path.insert(0, '.')
import numpy as np, torch
from gym_craftingworld_amd import CraftingWorldVecEnv
bad = 0
for name, N, T, kw in (('walker 8x8', 65536, 6000, dict(size=(8, 8), max_steps=300, reward_style='subset', selected_tasks=['EatBread'], number_of_tasks=1)),
                       ('max_steps 3, 5x5', 20000, 3000, dict(size=(5, 5), max_steps=3)),
                       ('walker 21x21 pixels', 16384, 2500, dict(size=(21, 21), max_steps=60, reward_style='subset', selected_tasks=['EatBread', 'GoToHouse'], number_of_tasks=1, obs_mode='pixels'))):
    acts = torch.randint(0, 4, (256, N), device='cuda', dtype=torch.uint8, generator=torch.Generator(device='cuda').manual_seed(3))
    kw.setdefault('obs_mode', 'state')
    os.environ.pop('CW_TUNE_LOOKAHEAD', None)
    a = CraftingWorldVecEnv(N, seed=5, **kw)
    os.environ['CW_TUNE_LOOKAHEAD'] = '0'
    b = CraftingWorldVecEnv(N, seed=5, **kw)
    os.environ.pop('CW_TUNE_LOOKAHEAD', None)
    assert a.tuner_state()['lookahead'] == 1 and b.tuner_state()['lookahead'] == 0
    a.reset(); b.reset()
    for t in range(T):
        a.step_async(acts[t % 256]); b.step_async(acts[t % 256])
        if t % 500 == 499:
            ok = torch.equal(a.hdr, b.hdr) and torch.equal(a.slot_pos, b.slot_pos) and torch.equal(a.reward, b.reward) and torch.equal(a.done, b.done)
            if kw['obs_mode'] == 'pixels':
                ok = ok and all(torch.equal(a._observation()[k], b._observation()[k]) for k in ('observation', 'desired_goal', 'init_observation'))
            bad += not ok
    (ka, pa), (kb, pb) = a.get_rng_states(), b.get_rng_states()
    ok = np.array_equal(ka, kb) and np.array_equal(pa, pb) and torch.equal(a.counters, b.counters)
    bad += not ok
    c = a._counters_raw.cpu()
    print('%-22s %d envs x %d steps: episodes %d, slow-path resets with the ring %d (without: every one), equal %s' % (name, N, T, int(c[1]), int(c[5]), ok and not bad), flush=True)
    a.close(); b.close()
print('bad', bad)
