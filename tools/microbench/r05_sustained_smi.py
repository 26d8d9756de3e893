"""GPU box: what changes on the card during a sustained run?  40 s of back-to-back steps of the headline batch (phases spread out, guard on, CW_TUNE_VERBOSE=1)
with `rocm-smi` read once a second beside it (clocks, power, temperatures), and the time per step of every 4 096 steps.
python tools/microbench/r05_sustained_smi.py [seconds]"""
import json
import subprocess
import sys
import threading
import time
import numpy as np
import torch
sys.path.insert(0, '.')
from gym_craftingworld_amd import CraftingWorldVecEnv
SECS = float(sys.argv[1]) if len(sys.argv) > 1 else 40.0
N = 65536
acts = torch.randint(0, 6, (512, N), device='cuda', dtype=torch.uint8, generator=torch.Generator(device='cuda').manual_seed(5))
env = CraftingWorldVecEnv(N, obs_mode='pixels', size=(21, 21), max_steps=300, seed=2024)
env.reset(); env.set_state(step_num=((np.arange(N) * 7) % 300).astype(np.int32))
stop = False
log = []


def smi():
    while not stop:
        t = time.perf_counter()
        try:
            out = subprocess.run(['rocm-smi', '-d', '0', '--showclocks', '--showpower', '--showtemp', '--json'], capture_output=True, text=True, timeout=10).stdout
            d = json.loads(out)
            c = d[next(iter(d))]
            log.append((t, {k: v for k, v in c.items() if any(w in k.lower() for w in ('sclk', 'mclk', 'fclk', 'socclk', 'power', 'temperature'))}))
        except Exception as exc:  # noqa: BLE001
            log.append((t, {'error': str(exc)[:100]}))
        time.sleep(1.0)


th = threading.Thread(target=smi, daemon=True)
th.start()
time.sleep(2.5)                                      # (the idle card first)
t0 = time.perf_counter()
n, marks = 0, []
while time.perf_counter() - t0 < SECS:
    for _ in range(4096):
        env.step_async(acts[n % 512])
        n += 1
    torch.cuda.synchronize()
    marks.append((time.perf_counter(), n, env.tuner_state()['period16']))
stop = True
th.join()
prev_t, prev_n = t0, 0
for t, k, p16 in marks:
    near = min(log, key=lambda e: abs(e[0] - t))[1] if log else {}
    print('%6.1f s  %.4f ms per step  clock %d  %s' % (t - t0, (t - prev_t) / (k - prev_n) * 1e3, p16, json.dumps(near)), flush=True)
    prev_t, prev_n = t, k
print('idle before the run:', json.dumps(log[0][1]) if log else None)
env.close()
