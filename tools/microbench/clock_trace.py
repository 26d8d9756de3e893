"""GPU box only: is the render kernel's launch time a function of TIME (power / clock management) rather than of the kernel?
Launches cw_render back to back for a few seconds, timing every launch with events, while a thread samples the card's clocks and power
from sysfs; then the same with idle gaps between launches.   python clock_trace.py [pace] [seconds]"""
import glob
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from gym_craftingworld_amd import CraftingWorldVecEnv  # noqa: E402

N = 65536
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0


def sysfs_files():
    out = {}
    for card in sorted(glob.glob('/sys/class/drm/card*/device')):
        for name in ('pp_dpm_sclk', 'pp_dpm_mclk', 'pp_dpm_fclk', 'pp_dpm_socclk', 'gpu_busy_percent', 'mem_busy_percent'):
            p = os.path.join(card, name)
            if os.path.exists(p):
                out.setdefault(card, {})[name] = p
        for hw in glob.glob(os.path.join(card, 'hwmon/hwmon*')):
            for name in ('power1_average', 'power1_input', 'temp1_input', 'temp2_input', 'temp3_input', 'freq1_input', 'freq2_input'):
                p = os.path.join(hw, name)
                if os.path.exists(p):
                    out.setdefault(card, {})[name] = p
    return out


def read_state(files):
    s = {}
    for name, p in files.items():
        try:
            txt = open(p).read().strip()
        except OSError:
            continue
        if name.startswith('pp_dpm'):
            cur = [ln for ln in txt.splitlines() if ln.endswith('*')]
            s[name[7:]] = cur[0].split(':')[1].strip(' *') if cur else txt.replace('\n', '|')
        else:
            s[name] = txt
    return s


cards = sysfs_files()
print('sysfs cards:', {c: sorted(f) for c, f in cards.items()})
env = CraftingWorldVecEnv(N, obs_mode='state', seed=0)
env.reset()
out = torch.empty((N, 84, 84, 3), dtype=torch.uint8, device='cuda')
samples = []
stop = False


def sampler():
    while not stop:
        t = time.monotonic()
        samples.append((t, {c.split('/')[4]: read_state(f) for c, f in cards.items()}))
        time.sleep(0.05)


def trace(label, gap_us):
    global stop, samples
    samples, stop = [], False
    th = threading.Thread(target=sampler)
    th.start()
    t0 = time.monotonic()
    series = []
    while time.monotonic() - t0 < seconds:
        evs = []
        for _ in range(100):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); env.render(out); b.record()
            evs.append((a, b))
            if gap_us:
                torch.cuda._sleep(int(gap_us * 2400))
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(b) for a, b in evs)
        series.append((time.monotonic() - t0, ms[50], ms[0], ms[-1]))
    stop = True
    th.join()
    print('== %s' % label)
    for t, med, mn, mx in series[:: max(1, len(series) // 40)]:
        near = min(samples, key=lambda s: abs(s[0] - t0 - t))[1]
        busy = {c: {k: v for k, v in st.items()} for c, st in near.items()}
        hot = max(busy.items(), key=lambda kv: float(kv[1].get('power1_average', kv[1].get('power1_input', '0')) or 0))
        print('t %.2f s  launch median %.4f ms (min %.4f max %.4f)   %s %s' % (t, med, mn, mx, hot[0], hot[1]))


trace('back to back', 0)
trace('100 us idle between launches', 100)
trace('back to back again', 0)
