import sys, time
sys.path.insert(0, '.')
import torch
from gym_craftingworld_amd import CraftingWorldVecEnv
N, T = 65536, 2000
acts = torch.randint(0, 4, (256, N), device='cuda', dtype=torch.uint8)
for mode, S, keep in (('pixels_dirty', 8, False), ('pixels_dirty', 8, True), ('pixels', 8, False), ('pixels', 8, True), ('pixels', 21, False), ('pixels', 21, True)):
    env = CraftingWorldVecEnv(N, obs_mode=mode, size=(S, S), max_steps=300, seed=1, reward_style='subset', selected_tasks=['EatBread'], number_of_tasks=1, keep_terminal_obs=keep)
    env.reset()
    for t in range(600): env.step_async(acts[t % 256])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in range(T): env.step_async(acts[t % 256])
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('%-12s S=%d keep_terminal=%d: %.2f us/step' % (mode, S, keep, dt / T * 1e6), flush=True)
    env.close()
