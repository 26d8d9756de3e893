#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the batched CraftingWorld step/reset hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    N>1 works both ways: under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...`
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the env), or started plainly -- then this process, before it touches
    torch or HIP, starts N fresh child ranks (gym_craftingworld_amd/launch.py), relays rank 0's JSON line and exits with
    the worst child's exit code.

Workload = BASELINE.json configs[2] (the config the metric is quoted on): 65 536 envs per GPU,
21x21 grid, full-frame 4x4-pixel-cell uint8 observation every step, auto-reset, uniform random
actions, max_steps=300.  One "step" = one cw_step over the whole batch: cw_step_fused_kernel (every env's step; a finished env takes
the look-ahead record of its next episode and its workgroup's four waves paint the two frames a reset changes), then cw_render_pieces_kernel, the clocked
sweep that writes the whole observation array; every max_steps/4-th step cw_refill_kernel ahead of them.  Envs shard across
ranks with no data-path collective (weak scaling: 65 536 envs per GPU); the only collectives
are the timing barrier and the max-over-ranks of the elapsed time.  `value` is computed from the slowest rank's DEVICE time for the K steps (two events
on the launch stream inside the barrier + synchronize bracket); `value_wall` from the wall clock around the bracket (it also holds the closing barrier).

Prints ONE JSON line on rank 0 (contract in the task prompt), with
  roofline     -- dominant kernel (cw_render_pieces_kernel<raster, frames per job>: the sweep of the observation array) vs the HBM roof, from
                  HIP events recorded by the library on the launch stream (cw_profile_begin/end) over a second, identical
                  K-step region (the first region, without events, gives `value`); step_frac = the WHOLE step against the same roof;
  policy_in_loop -- the same engine with a consumer between two steps that reads every observation byte and produces the next actions;
  cpu_baseline -- the CPU oracle (C port of the reference algorithm, oracle/) ALONE on this host's whole CPU share,
                  same workload shape, bounded sample; `value` is like for like with the headline (a full
                  render() per step), `dirty_cell_value` is the reference's own repaint strategy; `beside_gpu_soak`: a second run beside
                  soak_beside_cpu_baseline (the GPU kept busy meanwhile; never the baseline);
  short_episodes_1gpu -- what finishing episodes cost when a policy succeeds (episodes of ~140 steps under max_steps 300);
  metric_window -- SURVEY 8d's window whatever --steps says: 2*max_steps consecutive steps (both synchronized
                  time-out steps inside), timed in the same process, with the two slowest steps;
  repeats      -- the K-step region timed three times (value is the first, as the contract says);
  metric_window_desync -- the same window with the episode phases spread out (about N/max_steps envs finish on EVERY step: the steady
                  state of a run whose policy finishes episodes), same process, with its own roofline fraction;
  single_env   -- BASELINE configs[0]: make('craftingworld-v3'), 200 random steps through the N=1 facade;
  per_rank_ms_per_step -- every rank's own time for the timed region (gathered once, after it): stragglers show in a scaling run.
The JSON line is the LAST line of rank 0's stdout; everything else any library prints to fd 1 is sent to stderr.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


def host_cpu_share():
    """CPUs this process may actually use: the affinity mask capped by the cgroup CPU quota (a GPU box hands a
    16-CPU share to a container that still sees all 256 hardware threads)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:                                   # cgroup v2: "<quota|max> <period>"
            quota, period = f.read().split()[:2]
        if quota != 'max':
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        try:                                                                        # cgroup v1
            with open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us') as f:
                quota = int(f.read())
            with open('/sys/fs/cgroup/cpu/cpu.cfs_period_us') as f:
                period = int(f.read())
            if quota > 0 and period > 0:
                n = min(n, max(1, -(-quota // period)))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(size, max_steps, seconds=12.0, spare_cores=0):
    """Time the CPU oracle (port of the reference's step()/reset()) on this host's CPU share: n envs x T steps with
    auto-reset, sized to ~`seconds` of wall time.  Two rates: `value` with the whole frame rendered after every step -- the work
    the GPU headline configuration does -- and `dirty_cell_value` with the reference's own observation strategy (render_edit: the
    persistent frame, <= 2 cells repainted per step); beside them `single_env_one_core`, BASELINE configs[0] on one core.  spare_cores: threads left to the caller
    (bench.py runs this beside a GPU soak leg whose launching thread needs one: `cores` reports what the oracle used)."""
    from oracle import OracleBatch
    cores = max(1, host_cpu_share() - spare_cores)
    n = 64 * cores
    keys_pos = []
    for i in range(n):
        st = np.random.RandomState(9000 + i).get_state()
        keys_pos.append((st[1], st[2]))
    batch = OracleBatch(n, rng_states=keys_pos, size=(size, size), max_steps=max_steps)
    batch.reset()
    rng = np.random.RandomState(1)
    T0 = 2 * max_steps
    acts = rng.randint(0, 6, size=(T0, n)).astype(np.int8)

    def timed(fn, budget):
        t0 = time.perf_counter()
        fn(acts, nthreads=cores)                                 # calibration, also warms caches
        dt = time.perf_counter() - t0
        reps = max(1, int(budget / max(dt, 1e-3)))
        total, t0 = 0, time.perf_counter()
        for _ in range(reps):
            total += fn(acts, nthreads=cores)
        dt = time.perf_counter() - t0
        return total / dt, reps, dt

    rate_dirty, reps, dt = timed(batch.rollout, 0.4 * seconds)
    rate_full, reps_full, dt_full = timed(batch.rollout_full_frame, 0.6 * seconds)
    # BASELINE configs[0] on the CPU (SURVEY 8d ii): ONE env on ONE core, the reference's own loop shape -- 200 random steps, reset on done -- many times over
    one = OracleBatch(1, rng_states=keys_pos[:1], size=(size, size), max_steps=max_steps)
    one.reset()
    a1 = rng.randint(0, 6, size=(200, 1)).astype(np.int8)
    one.rollout(a1, nthreads=1)
    t0, n1 = time.perf_counter(), 0
    while time.perf_counter() - t0 < 0.25:
        n1 += one.rollout(a1, nthreads=1)
    single_core = n1 / (time.perf_counter() - t0)
    calib = None
    try:   # port vs the reference's own Python, both timed in the build container (tools/calibrate_cpu.py)
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tests', 'golden', 'cpu_calibration.json')) as f:
            c = json.load(f)
        calib = dict(port_over_reference_1core=c['port_over_reference'],
                     reference_python_env_steps_per_s_core=c['reference_python_env_steps_per_s'],
                     reference_equivalent_env_steps_per_s=rate_dirty / c['port_over_reference'],
                     note='ratio of the dirty-cell port to the reference (which repaints dirty cells), measured in the build '
                          'container on one core, not on this host; reference_equivalent = dirty_cell_value / ratio')
    except (OSError, KeyError, ValueError):
        pass
    return dict(value=rate_full, unit='env-steps/s', cores=cores, kind='port', dirty_cell_value=rate_dirty, calibration=calib,
                single_env_one_core={'value': single_core, 'unit': 'env-steps/s', 'workload': 'BASELINE configs[0]: one env, 200 random steps at a time, reset on done, '
                                     'one core, the reference\'s dirty-cell repaint (the reference\'s own Python: 64-78 k env-steps/s, BASELINE.md 2)'},
                like_for_like='value renders the whole frame after every step, as the GPU headline does; dirty_cell_value '
                              'keeps the reference\'s persistent frame and repaints <= 2 cells per step (compare with '
                              'other_obs_modes_1gpu.pixels_dirty)',
                reference_note='the reference itself (pure Python) cannot travel to the GPU box; BASELINE.md §2 has it at '
                               '64-78 k env-steps/s on one 2.1 GHz Xeon core (measured in the build container)',
                sample='%d envs x %d steps (%dx%d, max_steps=%d, auto-reset), %.1f s on %d OpenMP threads (%s) '
                       'with a full render() per step -> value; x %d steps, %.1f s with the reference\'s dirty-cell repaint -> '
                       'dirty_cell_value' % (n, T0 * reps_full, size, size, max_steps, dt_full, cores,
                                             'the whole cgroup CPU share, nothing else running in this process' if spare_cores == 0 else
                                             'the cgroup CPU share minus %d for the thread that keeps the GPU soak leg going beside it' % spare_cores,
                                             T0 * reps, dt))


def single_env_latency(device, steps=200, repeats=5):
    """BASELINE configs[0] through the product's N=1 facade: make('craftingworld-v3') (21x21, pixel Dict obs, numpy in/out,
    no auto-reset), `steps` uniform random actions, reset() on done -- the reference's own loop (gen_info.rst:62-82).
    The reference runs this at 16.9 us median per step on one 2.1 GHz core.  Measured twice: with the resident stepper
    (default: step() rings a doorbell a resident one-wave kernel polls) and with the launch path (resident=False: one kernel launch + one
    stream sync per step, rounds 1-2)."""
    import gym_craftingworld_amd as cw
    acts = np.random.RandomState(0).randint(0, 6, size=steps)
    out = {}
    for name, resident in (('resident', True), ('launch_path', False)):
        env = cw.make('craftingworld-v3', device=device, seed=0, resident=resident)
        env.reset()
        for a in acts[:50]:
            if env.step(a)[2]:
                env.reset()
        runs = []
        for _ in range(repeats):
            env.reset()
            t0 = time.perf_counter()
            for a in acts:
                if env.step(a)[2]:
                    env.reset()
            runs.append((time.perf_counter() - t0) / steps * 1e6)
        t0 = time.perf_counter()
        for _ in range(20):
            env.reset()
        reset_us = (time.perf_counter() - t0) / 20 * 1e6
        env.close()
        runs.sort()
        out[name] = dict(us_per_step=runs[len(runs) // 2], us_per_step_min=runs[0], us_per_step_runs=runs, us_per_reset=reset_us)
    return dict(workload="BASELINE configs[0]: make('craftingworld-v3'), %d random steps, numpy Dict obs on the host" % steps,
                steps=steps, us_per_step=out['resident']['us_per_step'], us_per_step_min=out['resident']['us_per_step_min'],
                us_per_step_runs=out['resident']['us_per_step_runs'], us_per_reset=out['resident']['us_per_reset'],
                us_per_step_launch_path=out['launch_path']['us_per_step'], us_per_reset_launch_path=out['launch_path']['us_per_reset'],
                reference_us_per_step=16.9, reference_us_per_reset=374.0,
                note='step(): a doorbell word in pinned host memory polled by a resident one-wave kernel (cw_step_resident), outputs and the '
                     '<= 2 repainted cells written straight into pinned host memory -- no launch, no stream sync; us_per_step_launch_path is '
                     'the rounds 1-2 path (one launch + one sync per step); reference_*: the reference\'s own Python on one 2.1-GHz core, '
                     'measured in the build container (BASELINE.md §2)')


def make_consumer(kind, n_envs, frame_shape, dev):
    """A stand-in POLICY for the policy-in-the-loop measurement (SURVEY 8b's callers: "torch policies consuming device tensors without
    host copies", docs/source/envs/gen_info.rst:62-82 with a network where the reference samples).  -> fn(obs uint8 [N, H, W, 3]) -> actions
    uint8 [N]; torch kernels on the current stream, every observation byte read, the next actions a function of it.
      reduce: obs.view(N, -1).sum(1) % 6 -- one pass over the frames, nothing else.  reduce32: the same pass over int32 words (15x faster).
      conv:   two strided convolutions (kernel = stride: 4x4 per cell -- or 3x3 for the AltObs tiles -- then 3x3 cells), bf16, as GEMMs over
              the patches, ReLU between, mean over positions, a 6-way head, an action sampled from its softmax (Gumbel-max).  Weights are fixed random numbers (seed 0)."""
    import torch
    H, Wd, _ = frame_shape
    if kind == 'reduce':
        def policy(obs):
            return torch.remainder(obs.view(n_envs, -1).sum(1, dtype=torch.int32), 6).to(torch.uint8)
        policy.describe = 'obs.view(N,-1).sum(1) % 6 (torch reduction over every observation byte; torch sums uint8 at ~0.4 TB/s)'
        return policy
    if kind == 'reduce32':
        # the same bytes read as int32 words: torch's fastest reduction (6.2 TB/s, tools/microbench/consumer_speed.py) -- a READER at the
        # memory's own pace, so that sweep and reader alternate at full rate (frames are multiples of 16 bytes for the Ray raster; AltObs
        # frames are not multiples of 4: the byte-wise sum there)
        if (H * Wd * 3) % 4:
            return make_consumer('reduce', n_envs, frame_shape, dev)

        def policy(obs):
            return torch.remainder(obs.view(n_envs, -1).view(torch.int32).sum(1, dtype=torch.int32), 6).to(torch.uint8)
        policy.describe = 'obs.view(N,-1).view(int32).sum(1) % 6 (every observation byte, as int32 words: torch\'s fastest reader)'
        return policy
    if kind != 'conv':
        raise ValueError('consumer must be reduce, reduce32 or conv')
    p1 = 4 if H == Wd else 3                                   # one cell per patch (Ray 4x4 px, AltObs 3x3 px; its strip of 3 more rows is one more patch row)
    g1h, g1w = H // p1, Wd // p1
    p2 = 3
    g2h, g2w = g1h // p2, g1w // p2
    gen = torch.Generator(device=dev).manual_seed(0)
    w1 = (torch.randn(p1 * p1 * 3, 16, device=dev, generator=gen) / 64.0).to(torch.bfloat16)
    w2 = (torch.randn(p2 * p2 * 16, 32, device=dev, generator=gen) / 12.0).to(torch.bfloat16)
    w3 = torch.randn(32, 6, device=dev, generator=gen).to(torch.bfloat16)
    chunk = 8192                                               # envs per pass: the bf16 patches of a chunk stay a few hundred MB

    def policy(obs):
        out = torch.empty(n_envs, dtype=torch.uint8, device=dev)
        for lo_ in range(0, n_envs, chunk):
            o = obs[lo_:lo_ + chunk]
            n = o.shape[0]
            x = o[:, :g1h * p1, :g1w * p1].reshape(n, g1h, p1, g1w, p1, 3).permute(0, 1, 3, 2, 4, 5).reshape(n * g1h * g1w, p1 * p1 * 3)
            h1 = torch.relu(x.to(torch.bfloat16) @ w1).view(n, g1h, g1w, 16)
            y = h1[:, :g2h * p2, :g2w * p2].reshape(n, g2h, p2, g2w, p2, 16).permute(0, 1, 3, 2, 4, 5).reshape(n * g2h * g2w, p2 * p2 * 16)
            h2 = torch.relu(y @ w2).view(n, g2h * g2w, 32).mean(1)
            logits = (h2 @ w3).float()
            logits = logits / (logits.abs().amax(1, keepdim=True) + 1.0)          # (fixed random weights: keep the head's temperature near 1)
            u = torch.rand(logits.shape, device=dev, generator=gen).clamp_(1e-9, 1.0 - 1e-7)
            out[lo_:lo_ + chunk] = (logits - torch.log(-torch.log(u))).argmax(1).to(torch.uint8)      # a SAMPLED action (Gumbel-max), as a stochastic policy takes it
        return out
    policy.describe = ('conv %dx%d/%d (3->16) + ReLU, conv 3x3/3 (16->32) + ReLU, mean, linear 32->6, sampled action; bf16 GEMMs over patches, %d envs per pass'
                       % (p1, p1, p1, chunk))
    return policy


def frames_per_job(frame_bytes):
    """template parameter of cw_render_pieces_kernel as a kernel trace lists it: the most frames an aligned 4-KiB piece overlaps, as a power of two"""
    most, fpj = 4095 // frame_bytes + 2, 2
    while fpj < most:
        fpj <<= 1
    return fpj


def place_short_region(prewarm, warmup, steps, max_steps):
    """-> the number of untimed device warm-up steps to take so that a timed region SHORTER than an episode holds none of the steps on which
    every env times out at once (every max_steps-th step with synchronized phases, ~4x a plain step).  Such a region (the driver's K may be 20
    steps) would hold one of them or none -- 1 in 20 instead of 1 in 300, +15 % either way; metric_window is the figure that includes them in
    their true proportion.  Regions of max_steps steps or more are left where they are."""
    if steps >= max_steps:
        return prewarm
    first = prewarm + warmup + 1                         # 1-based number of the first timed step since reset()
    if (first - 1) // max_steps != (first + steps - 1) // max_steps:
        prewarm += max_steps - (first - 1) % max_steps + min(8, max_steps - steps - 1)   # just past the time-out step (a few steps of slack if they fit)
    return prewarm


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=600)      # >= 2*max_steps: both synchronized time-out steps are inside
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--envs-per-gpu', type=int, default=65536)
    ap.add_argument('--size', type=int, default=21)
    ap.add_argument('--max-steps', type=int, default=300)
    ap.add_argument('--obs-mode', default='pixels', choices=['pixels', 'pixels_dirty', 'state'])
    ap.add_argument('--raster', default='ray', choices=['ray', 'alt'], help="'alt' = CraftingWorldEnvAltObs tiles (side measurement)")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-seconds', type=float, default=12.0)
    ap.add_argument('--dist-backend', default='nccl', choices=['nccl', 'gloo'],
                    help='nccl (= RCCL) in production; gloo only to rehearse the multi-rank flow')
    ap.add_argument('--rehearse-on-one-gpu', action='store_true',
                    help='map every rank to cuda:0 (multi-rank control-flow rehearsal on a 1-GPU box; use with gloo)')
    ap.add_argument('--rollout', action='store_true',
                    help='state-only mode: run the K steps as ONE persistent-kernel launch (cw_rollout; actions known up front)')
    ap.add_argument('--no-other-modes', action='store_true', help='skip the short dirty-cell / state-only side measurements')
    ap.add_argument('--desync', action='store_true',
                    help='spread the episode phases out after the reset (step_num = 7 e mod max_steps): about N/max_steps envs '
                         'finish on EVERY step, the steady state of a long run (default: all envs start together, so the window '
                         'holds two steps on which every env times out at once, as SURVEY 8d defines the metric)')
    ap.add_argument('--graph-steps', type=int, default=0,
                    help='capture this many consecutive steps into one HIP graph and replay it (0 = eager launches)')
    ap.add_argument('--mixed-menus', action='store_true',
                    help='BASELINE configs[3] shape: env i uses ordered task list i mod 8 of a fixed menu of eight (heterogeneous selected_tasks / '
                         'number_of_tasks / stacking / reward_style per env)')
    ap.add_argument('--prewarm-steps', type=int, default=100,
                    help='untimed steps before the W warm-up steps (0: none): a card that idled through set-up runs its first ~100 launches 3-5 %% '
                         'slower; reported as prewarm_steps / warmup_total')
    ap.add_argument('--launch-timeout', type=float, default=1500.0, help='self-launched ranks are stopped after this many seconds')
    ap.add_argument('--no-single-env', action='store_true', help='skip the N=1 facade latency (BASELINE configs[0])')
    ap.add_argument('--consumer', default='default', choices=['default', 'none', 'reduce', 'reduce32', 'conv', 'both', 'all'],
                    help='policy-in-the-loop measurement (pixel modes): between two steps a torch kernel that reads EVERY observation byte and '
                         'produces the next actions from it, on the env\'s stream.  reduce: obs.view(N,-1).sum(1) %% 6; reduce32: the same over int32 words '
                         '(a reader at the memory\'s pace); conv: two strided bf16 convolutions + a sampled action.  default: all three, except with '
                         '--quick (none); a named one also runs with --quick')
    ap.add_argument('--consumer-steps', type=int, default=0, help='steps per pass of the policy-in-the-loop measurement (0: 2*max_steps)')
    ap.add_argument('--shard-check', type=int, default=0, metavar='T',
                    help='no timing: reset, take T steps with actions that depend only on (step, GLOBAL env index), write per-env CRC32s of '
                         'the final frames and the reward / done of every step of this rank\'s shard to --shard-out/rank<r>.npz and exit '
                         '(tests: shards of self-launched ranks == the single-batch run)')
    ap.add_argument('--shard-out', default='gpurun_out/shard_check')
    ap.add_argument('--inject-sleep-ms', type=float, default=0.0,
                    help='test hook: the rank named by --inject-sleep-rank sleeps this long on the HOST between its last launch of a timed region and the '
                         'closing barrier (a late rank at the barrier must show in value_wall only, not in value: the region is timed on the device)')
    ap.add_argument('--inject-sleep-rank', type=int, default=1)
    ap.add_argument('--quick', action='store_true',
                    help='profiling runs (rocprofv3 / PMC passes serialise kernels): only the contract regions -- no repeats, no metric '
                         'window, no other modes, no single env, no CPU baseline')
    args = ap.parse_args()
    if args.quick:
        args.no_cpu_baseline = args.no_other_modes = args.no_single_env = True

    # Plain `python bench.py --gpus N`: become the parent of N fresh ranks BEFORE anything touches torch or HIP (a
    # process that has initialised the GPU must never be replaced or forked on this pool).  launch.py is loaded by
    # path so that not even the package (which imports torch) is imported here.
    import importlib.util
    spec = importlib.util.spec_from_file_location('cw_launch', os.path.join(ROOT, 'gym_craftingworld_amd', 'launch.py'))
    launch = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(launch)
    if launch.needs_self_launch(args.gpus):
        sys.exit(launch.self_launch(args.gpus, os.path.abspath(__file__), sys.argv[1:], timeout=args.launch_timeout))

    # The JSON line must be the LAST line of rank 0's stdout, and libraries write to fd 1 as they please ("[Gloo] Rank 0 is connected to
    # ..."): keep the real stdout aside for that one line and point fd 1 at stderr for everything else.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), 'w')
    os.dup2(2, 1)

    # one process per GPU: stay on the CPUs of the NUMA node that GPU hangs off (before anything touches the GPU; multi-rank runs only -- a single
    # rank keeps the whole CPU share for the CPU baseline)
    numa = launch.bind_to_gpu_numa(int(os.environ.get('LOCAL_RANK', '0'))) if int(os.environ.get('WORLD_SIZE', '1')) > 1 and not args.rehearse_on_one_gpu \
        else {'skipped': 'single rank' if int(os.environ.get('WORLD_SIZE', '1')) == 1 else 'rehearsal: every rank on GPU 0'}

    import torch
    import torch.distributed as dist

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit('WORLD_SIZE=%d does not match --gpus %d' % (world, args.gpus))
    if args.rehearse_on_one_gpu:
        local_rank = 0
    masked = [v for v in ('HIP_VISIBLE_DEVICES', 'ROCR_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES') if os.environ.get(v)]
    if world > 1 and masked and torch.cuda.device_count() == 1:
        local_rank = 0                                   # (a launcher that hands every rank ITS GPU through a *_VISIBLE_DEVICES mask: that GPU is device 0 here)
    if local_rank >= torch.cuda.device_count():
        raise SystemExit('rank %d needs GPU %d but only %d are visible (use --rehearse-on-one-gpu with --dist-backend gloo '
                         'to rehearse the multi-rank flow on one GPU)' % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    backend_used = None
    timing_group = None          # the group the timing barrier and the max-over-ranks run on (None = the default group)
    if world > 1:
        import datetime
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        # The default group is gloo (CPU, rendezvous over MASTER_ADDR:MASTER_PORT): it carries the control plane -- above all the agreement
        # on whether RCCL works.  RCCL (backend "nccl") is a second group for the two timing collectives; no data-path collective exists.
        # sharding.agree_on_rccl: every rank publishes its status over gloo after creating the group and again after one all-reduce on it --
        # everybody uses RCCL or nobody does, whatever subset of ranks saw a failure.  The RCCL group waits in blocking mode
        # (TORCH_NCCL_BLOCKING_WAIT=1 while it is created, restored afterwards): a rank whose peers never arrive in the probe gets an exception after
        # the group's 60-s timeout and joins the agreement; a timing collective that hangs later raises from its wait() after the same timeout (blocking
        # wait switches the watchdog's asynchronous error handling off for this group), the rank ends non-zero and launch.py stops the others.
        dist.init_process_group('gloo', timeout=datetime.timedelta(seconds=300))
        backend_used = args.dist_backend
        if args.dist_backend == 'nccl':
            from gym_craftingworld_amd.sharding import agree_on_rccl
            timing_group, why = agree_on_rccl(dev, log=lambda m: sys.stderr.write('[bench] %s\n' % m))
            if timing_group is None:
                args.dist_backend = 'gloo'
                backend_used = 'gloo (RCCL group failed on at least one rank: %s)' % why

    from gym_craftingworld_amd import CraftingWorldVecEnv
    from gym_craftingworld_amd.sharding import gather_objects_over_ranks, gather_over_ranks, max_over_ranks, shard_range

    K, W = args.steps, args.warmup
    # weak scaling: the global batch is envs_per_gpu * world envs; rank g owns the contiguous range
    # [lo, hi) and env e's stream is numpy RandomState(e) whatever rank owns it
    lo, hi = shard_range(rank, world, args.envs_per_gpu * world)
    N = hi - lo
    menu_kw = {}
    if args.mixed_menus:
        T = ['MakeBread', 'EatBread', 'BuildHouse', 'ChopTree', 'ChopRock', 'GoToHouse', 'MoveAxe', 'MoveHammer', 'MoveSticks']
        menu_kw = dict(task_menus=[dict(), dict(selected_tasks=T[::-1]), dict(selected_tasks=T[:4], number_of_tasks=2),
                                   dict(selected_tasks=['GoToHouse', 'MoveAxe', 'EatBread'], stacking=False),
                                   dict(selected_tasks=T[3:], reward_style='subset'), dict(selected_tasks=['ChopTree', 'BuildHouse'], number_of_tasks=1),
                                   dict(selected_tasks=T[1::2]), dict(selected_tasks=T[::2], number_of_tasks=3, reward_style='subset')],
                       env_menu=(np.arange(lo, hi) % 8).astype(np.uint8))
    env = CraftingWorldVecEnv(N, size=(args.size, args.size), max_steps=args.max_steps, obs_mode=args.obs_mode,
                              device=dev, seed=lo, raster=args.raster, **menu_kw)
    render_kernel = env.render_kernel_name()     # what a rocprofv3 kernel trace of this run lists the bracketed kernel as
    env.reset()
    if args.shard_check > 0:
        # Engine-level shard equivalence across PROCESSES: actions depend on (step, global env index) only, so the ranks of any sharding
        # must reproduce, env by env, what one process stepping the whole batch produces.  No timing.
        import zlib
        Tn = args.shard_check
        rews, dones = [], []
        for t in range(Tn):
            g = np.arange(lo, hi, dtype=np.uint64)
            a = (((g * np.uint64(2654435761) + np.uint64(t) * np.uint64(40503)) >> np.uint64(7)) % np.uint64(6)).astype(np.uint8)
            _, r, d, _ = env.step(torch.from_numpy(a).to(dev))
            rews.append(r.cpu().numpy().copy())
            dones.append(d.cpu().numpy().copy())
        env.synchronize()
        out = dict(lo=lo, hi=hi, reward=np.stack(rews), done=np.stack(dones), hdr=env.hdr.cpu().numpy(), counters=env.counters.cpu().numpy())
        if args.obs_mode != 'state':
            for key, t_ in (('obs_crc', env._obs), ('goal_crc', env._desired_img), ('init_crc', env._init_img)):
                h = t_.cpu().numpy()
                out[key] = np.array([zlib.crc32(h[i].tobytes()) for i in range(N)], dtype=np.uint32)
        keys, pos = env.get_rng_states()
        out['rng_crc'] = np.array([zlib.crc32(keys[i, 1:].tobytes()) ^ int(pos[i]) for i in range(N)], dtype=np.uint32)
        os.makedirs(args.shard_out, exist_ok=True)
        np.savez(os.path.join(args.shard_out, 'rank%d.npz' % rank), **out)
        env.close()
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            json_out.write(json.dumps({'shard_check': Tn, 'n_gpus': world, 'envs': [lo, hi], 'out': args.shard_out}) + '\n')
            json_out.flush()
        return
    if args.desync:
        env.set_state(step_num=((np.arange(lo, hi) * 7) % args.max_steps).astype(np.int32))
    # synthetic actions: uniform in [0,6), pre-generated on device, one row per step (not part of the env)
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    rows = min(max(K + W, 2 * args.max_steps), 1024)
    actions = torch.randint(0, 6, (rows, N), device=dev, dtype=torch.uint8, generator=gen)

    def barrier():
        if world > 1:
            dist.barrier(group=timing_group)
        torch.cuda.synchronize(dev)

    G = args.graph_steps
    graph = None
    if G > 0:
        if K % G or W % G or rows % G:
            raise SystemExit('--graph-steps must divide --steps, --warmup and the action-row pool')
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):                    # torch's capture warm-up protocol
            for t in range(G):
                env.step_async(actions[t])
        torch.cuda.current_stream(dev).wait_stream(side)
        env.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):                    # G steps, each reading its own action row
            for t in range(G):
                env.step_async(actions[t])

    if args.rollout and args.obs_mode != 'state':
        raise SystemExit('--rollout needs --obs-mode state')

    def run(k, t_off):
        if args.rollout:
            idx = (torch.arange(k, device=dev) + t_off) % rows
            env.rollout(actions[idx] if k != rows or t_off else actions, record=False)
            return
        if graph is not None:
            for _ in range(k // G):
                graph.replay()
            return
        for t in range(k):
            env.step_async(actions[(t_off + t) % rows])

    # Device warm-up before the contract's W warm-up steps: the card has idled through env construction, and its first few hundred
    # launches after that run 3-6 % slower than the steady state the K timed steps are meant to show (K=20, W=5 without it: 2.60-2.67 x 10^8
    # in the first region, 2.74-2.77 in the next two; profiles/history/r02_pace.txt T).  Untimed, like the W steps that follow it.
    prewarm_steps = 0 if args.rollout else max(args.prewarm_steps, 0)
    if prewarm_steps > 0 and not args.desync:
        prewarm_steps = place_short_region(prewarm_steps, W, K, args.max_steps)
    if prewarm_steps > 0:
        run((prewarm_steps // G) * G if G > 0 else prewarm_steps, 0)
        torch.cuda.synchronize(dev)
    run(W, 0)
    red_dev = dev if args.dist_backend == 'nccl' else 'cpu'

    # The timed region: barrier + synchronize on both sides, as the contract says -- and INSIDE it two events on the launch stream, one right after
    # the opening barrier (the stream is idle then), one behind the last step's launches.  `value` is computed from the slowest rank's DEVICE span
    # between the two: the k steps themselves.  The wall clock around the whole bracket also holds the closing barrier (100-200 us of RCCL at N > 1, nothing
    # at N = 1: at the driver's K = 20 that is 2-5 % of a 4-ms region read as scaling loss) and the host's wake-up after the last kernel; it is kept
    # beside it as value_wall.  SURVEY 8e: the scaling limit of this path is host launch overhead, not the interconnect -- so the barrier is not measured.
    def timed_region(k, t_off):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        barrier()
        t0 = time.perf_counter()
        e0.record()
        run(k, t_off)
        e1.record()
        if args.inject_sleep_ms > 0 and rank == args.inject_sleep_rank:
            time.sleep(args.inject_sleep_ms * 1e-3)
        barrier()
        timed_region.mine_wall = time.perf_counter() - t0
        timed_region.mine = e0.elapsed_time(e1) * 1e-3
        timed_region.wall = max_over_ranks(timed_region.mine_wall, device=red_dev, group=timing_group)
        return max_over_ranks(timed_region.mine, device=red_dev, group=timing_group)

    elapsed = timed_region(K, W)                     # THE timed region of the contract: exactly K steps after W warm-up steps
    elapsed_wall = timed_region.wall
    timed_region_first = (timed_region.mine, timed_region.mine_wall)
    per_rank_s = gather_over_ranks(timed_region.mine, device=red_dev, group=timing_group)      # (once, after the region: who was the slowest?)
    per_rank_wall_s = gather_over_ranks(timed_region.mine_wall, device=red_dev, group=timing_group)
    barrier()                                        # what an empty barrier pair costs this rank (control plane; explains value_wall - value at N > 1)
    tb = time.perf_counter()
    barrier()
    barrier_ms = (time.perf_counter() - tb) * 1e3
    # ... and twice more (the driver's K may be tiny: 20 steps are 5 ms); `value` stays the first region
    repeats_s, repeats_wall_s = [elapsed], [elapsed_wall]
    for r in range(0 if args.quick else 2):
        repeats_s.append(timed_region(K, W + (r + 1) * K))
        repeats_wall_s.append(timed_region.wall)
    t_next = W + 3 * K

    # second, identical K-step region with the library's HIP events around each kernel (eager launches:
    # events cannot be re-recorded from inside a replayed graph)
    launch_desc = ('persistent launches of max_steps steps for the K steps (cw_rollout)' if args.rollout else
                   ('eager' if G == 0 else 'hip graph of %d steps' % G))
    graph = None
    args.rollout = False
    episodes_before_prof = int(env.counters[1].item())
    env.profile_begin(K)
    barrier()
    t1 = time.perf_counter()
    run(K, t_next)
    barrier()
    elapsed_prof = time.perf_counter() - t1
    prof = env.profile_end()
    tuner = env.tuner_state()                        # (what pace the profiled launches ran at)
    t_next += K
    # every rank's own clock, guard moves, sweep time and NUMA binding (control plane, after the timed regions): a straggler in a scaling run explains itself
    S_r = args.size
    own_bytes = float(N) * (S_r * S_r + (48 * S_r * S_r if args.raster == 'ray' else 27 * S_r * (S_r + 1)))
    per_rank = gather_objects_over_ranks({
        'rank': rank, 'device': local_rank, 'envs': [lo, hi], 'tuner': tuner, 'numa': numa,
        'sweep_ms': prof['ms_render_kernel'] or None, 'sweep_ms_median': prof['ms_render_kernel_median'] or None,
        'roofline_frac': (own_bytes / (prof['ms_render_kernel'] * 1e-3) / 1e9 / HBM_PEAK_GBS) if args.obs_mode == 'pixels' and prof['ms_render_kernel'] > 0 else None,
        'ms_per_step_with_events': elapsed_prof / K * 1e3, 'barrier_ms': barrier_ms,
        'region_ms_device': timed_region_first[0] * 1e3, 'region_ms_wall': timed_region_first[1] * 1e3})
    episodes = int(env.counters[1].item())
    resets_in_prof = episodes - episodes_before_prof      # envs reset (and repainted: 3 frames each) inside the profiled launches

    # SURVEY 8d's metric window, whatever --steps was: 2*max_steps consecutive steps, so both steps on which (nearly)
    # every env times out at once are inside.  One clean pass for the rate; a second one with an event after every step
    # (on the stream the step joins back into) for the per-step durations.
    window = None
    if not args.quick:
        KW_ = 2 * args.max_steps
        ep0 = int(env.counters[1].item())
        win_elapsed = timed_region(KW_, t_next)
        win_wall = timed_region.wall
        win_episodes = int(env.counters[1].item()) - ep0
        t_next += KW_
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(KW_ + 1)]
        env.profile_begin(KW_)
        barrier()
        evs[0].record()
        for t in range(KW_):
            env.step_async(actions[(t_next + t) % rows])
            evs[t + 1].record()
        barrier()
        win_prof = env.profile_end()
        t_next += KW_
        step_ms = sorted((evs[t].elapsed_time(evs[t + 1]) for t in range(KW_)), reverse=True)
        window = {'steps': KW_, 'value': float(N) * world * KW_ / win_elapsed, 'unit': 'env-steps/s',
                  'ms_per_step': win_elapsed / KW_ * 1e3, 'value_wall': float(N) * world * KW_ / win_wall, 'episodes_finished': win_episodes,
                  'slowest_step_ms': step_ms[:2], 'median_step_ms': step_ms[KW_ // 2],
                  'render_kernel_ms_avg': win_prof['ms_render_kernel'] or None,
                  'launch': 'eager',
                  'note': 'same process, after the K-step regions; value from a pass without events (max over ranks), per-step '
                          'figures from a second pass with one event per step (rank 0)'}

    # ... and the same window with the episode phases spread out (ray.py:367: episodes end at different steps once a policy succeeds;
    # about N/max_steps envs finish on every step): the steady state of a long run, in the same process.
    window_desync = None
    if not args.quick and args.obs_mode == 'pixels' and not args.desync:
        KW_ = 2 * args.max_steps
        env.set_state(step_num=((np.arange(lo, hi) * 7) % args.max_steps).astype(np.int32))
        run(KW_, t_next)                                 # 2*max_steps untimed steps: every env has finished at its own phase, the look-ahead refills are in their rhythm
        t_next += KW_
        torch.cuda.synchronize(dev)
        ep0 = int(env.counters[1].item())
        wd_elapsed = timed_region(KW_, t_next)
        wd_wall = timed_region.wall
        wd_episodes = int(env.counters[1].item()) - ep0
        t_next += KW_
        env.profile_begin(KW_)
        barrier()
        run(KW_, t_next)
        barrier()
        wd_prof = env.profile_end()
        t_next += KW_
        wd_resets = int(env.counters[1].item()) - ep0 - wd_episodes
        S_ = args.size
        frame_ = 48 * S_ * S_ if args.raster == 'ray' else 27 * S_ * (S_ + 1)
        wd_bytes = float(N) * (S_ * S_ + frame_)        # (the bracketed kernel is the sweep of the observation array alone)
        wd_ms = wd_prof['ms_render_kernel']
        window_desync = {'steps': KW_, 'value': float(N) * world * KW_ / wd_elapsed, 'unit': 'env-steps/s', 'ms_per_step': wd_elapsed / KW_ * 1e3,
                         'value_wall': float(N) * world * KW_ / wd_wall,
                         'episodes_finished': wd_episodes, 'resets_per_step': wd_episodes / KW_,
                         'roofline': {'kernel': render_kernel, 'avg_launch_ms': wd_ms, 'median_launch_ms': wd_prof['ms_render_kernel_median'] or None,
                                      'achieved': wd_bytes / (wd_ms * 1e-3) / 1e9 if wd_ms > 0 else None,
                                      'frac': wd_bytes / (wd_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if wd_ms > 0 else None,
                                      'algorithmic_bytes_per_launch': wd_bytes},
                         'note': 'same process, after metric_window: step_num := 7 e mod max_steps, %d untimed steps, '
                                 'then the window without events (value, max over ranks) and once more with the library\'s events '
                                 '(roofline, rank 0)' % KW_}

    # POLICY IN THE LOOP (SURVEY 8b's callers: torch policies consuming device tensors without host copies; gen_info.rst:62-82 with a network
    # where the reference samples its actions).  Every region above runs sweeps back to back with nothing else on the card -- the regime the
    # sweep's clock, its head and the guard were tuned in.  Here a consumer that READS every byte of the 1.4 GB of frames runs between two steps
    # and its output is the next step's actions: step -> sweep -> consumer -> step ..., all on one stream.
    policy_blocks = None
    kinds = {'default': () if args.quick else ('reduce32', 'reduce', 'conv'), 'none': (), 'both': ('reduce', 'conv'),
             'all': ('reduce32', 'reduce', 'conv')}.get(args.consumer, (args.consumer,))
    if kinds and args.obs_mode == 'pixels':
        S_ = args.size
        frame_ = 48 * S_ * S_ if args.raster == 'ray' else 27 * S_ * (S_ + 1)
        sweep_bytes = float(N) * (S_ * S_ + frame_)
        policy_blocks = {}
        for kind in kinds:
            # (a pass of the conv consumer is 8.5 ms per step: a third of the window is plenty to time a 0.2-ms sweep 200 times)
            KC = args.consumer_steps if args.consumer_steps > 0 else (max(100, 2 * args.max_steps // 3) if kind == 'conv' else 2 * args.max_steps)
            policy = make_consumer(kind, N, env.frame_shape, dev)
            obs_t = env._obs
            a = actions[0].clone()
            for _ in range(3):                           # (the consumer's own warm-up: GEMM heuristics, allocator)
                a = policy(obs_t)
            torch.cuda.synchronize(dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                a = policy(obs_t)
            e1.record()
            e1.synchronize()
            consumer_alone_ms = e0.elapsed_time(e1) / 20
            tuner0 = env.tuner_state()
            for _ in range(64):
                env.step_async(a)
                a = policy(obs_t)
            barrier()
            ep0 = int(env.counters[1].item())
            t0 = time.perf_counter()                     # pass A: no events at all (the guard watches every 64th sweep)
            for _ in range(KC):
                env.step_async(a)
                a = policy(obs_t)
            barrier()
            wall = max_over_ranks(time.perf_counter() - t0, device=red_dev, group=timing_group)
            finished = int(env.counters[1].item()) - ep0
            tuner1 = env.tuner_state()
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(2 * KC)]
            env.profile_begin(KC)                        # pass B: the library's events around the sweep, torch events around the consumer
            barrier()
            t0 = time.perf_counter()
            for t in range(KC):
                env.step_async(a)
                evs[2 * t].record()
                a = policy(obs_t)
                evs[2 * t + 1].record()
            barrier()
            wall_b = time.perf_counter() - t0
            pp = env.profile_end()
            cons = sorted(evs[2 * t].elapsed_time(evs[2 * t + 1]) for t in range(KC))
            cons_ms = sum(cons) / KC
            ms_ = pp['ms_render_kernel']
            step_ms = wall / KC * 1e3
            policy_blocks[kind] = {
                'consumer': policy.describe, 'steps': KC, 'episodes_finished_per_step': finished / KC,
                'ms_per_step': step_ms, 'value': float(N) * world * KC / wall, 'unit': 'env-steps/s (consumer included)',
                'consumer_ms': cons_ms, 'consumer_ms_median': cons[KC // 2], 'consumer_alone_ms': consumer_alone_ms,
                'consumer_read_GBs': float(N) * frame_ / (cons_ms * 1e-3) / 1e9,
                'env_ms_per_step': step_ms - cons_ms,    # the whole step minus the consumer: step kernel + sweep + gaps
                'env_value': float(N) * world / ((step_ms - cons_ms) * 1e-3) if step_ms > cons_ms else None,
                'sweep': {'kernel': render_kernel, 'avg_launch_ms': ms_, 'median_launch_ms': pp['ms_render_kernel_median'] or None,
                          'launch_ms_min_max': [pp['ms_render_kernel_min'], pp['ms_render_kernel_max']],
                          'achieved': sweep_bytes / (ms_ * 1e-3) / 1e9 if ms_ > 0 else None,
                          'frac': sweep_bytes / (ms_ * 1e-3) / 1e9 / HBM_PEAK_GBS if ms_ > 0 else None,
                          'frac_at_median_launch': (sweep_bytes / (pp['ms_render_kernel_median'] * 1e-3) / 1e9 / HBM_PEAK_GBS
                                                    if pp['ms_render_kernel_median'] > 0 else None)},
                'step_frac': float(N) * (48 + S_ * S_ + frame_) / ((step_ms - cons_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS if step_ms > cons_ms else None,
                'ms_per_step_with_events': wall_b / KC * 1e3,
                'tuner_before': tuner0, 'tuner_after': tuner1,
                'guard_moves': tuner1['guard_slowdowns'] - tuner0['guard_slowdowns'],
                'note': 'step -> sweep -> consumer -> step on one stream; the consumer reads every observation byte and its output is the next '
                        'step\'s action tensor.  ms_per_step / value: %d steps without events (guard on); sweep / consumer_ms: the same again with the '
                        'library\'s events around the sweep and torch events around the consumer; episode phases as the regions above left '
                        'them (%s)' % (KC, 'spread out' if (window_desync or args.desync) else 'in step')}

    # THE CPU BASELINE -- ALONE: the oracle on the whole CPU share with this process doing nothing else (round 5 timed it beside the soak's Python launch
    # loop, which contends for the GIL around every ctypes call and for memory bandwidth: 6.7-6.9 M env-steps/s on 15 threads against 7.7-8.0 M on 16
    # in round 4 -- a GPU/CPU ratio inflated by 13 %).  `cpu_baseline.value` is this uncontended figure.
    # Then a GPU SOAK beside a SECOND, shorter oracle run (reported as cpu_baseline.beside_gpu_soak, never as the baseline): ~6 s of consecutive steps of the
    # headline batch, episode phases as the regions above left them, the clock's guard on -- the tuned constants outside a 600-step window, and a busy host
    # beside the card as a training loop would have it.
    cpu_result, soak = None, None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import threading
        torch.cuda.synchronize(dev)
        cpu_result = cpu_baseline(args.size, args.max_steps, args.cpu_seconds, spare_cores=0)
        box = {}

        def _cpu():
            try:
                box['r'] = cpu_baseline(args.size, args.max_steps, max(0.5 * args.cpu_seconds, 0.25), spare_cores=1)
            except Exception as exc:  # noqa: BLE001
                box['e'] = exc
        th = threading.Thread(target=_cpu, daemon=True)
        tuner0 = env.tuner_state()
        ep0 = int(env.counters[1].item())
        torch.cuda.synchronize(dev)
        th.start()
        t0 = time.perf_counter()
        n_soak, marks = 0, []
        while th.is_alive():
            for _ in range(256):
                env.step_async(actions[(t_next + n_soak) % rows])
                n_soak += 1
            if n_soak % 4096 == 0:                       # (keep the queue a few thousand steps deep at most; a mark every 4 096 steps)
                torch.cuda.synchronize(dev)
                marks.append(time.perf_counter() - t0)
        torch.cuda.synchronize(dev)
        soak_s = time.perf_counter() - t0
        th.join()
        if 'e' in box:
            raise box['e']
        cpu_result['beside_gpu_soak'] = {k: box['r'][k] for k in ('value', 'dirty_cell_value', 'cores', 'sample')}
        cpu_result['beside_gpu_soak']['single_env_one_core'] = box['r']['single_env_one_core']['value']
        cpu_result['beside_gpu_soak']['note'] = ('the same oracle run again while this process keeps the GPU soak leg going (a Python launch loop: GIL and memory '
                                                 'bandwidth shared): NOT the baseline')
        t_next += n_soak
        tuner1 = env.tuner_state()
        if args.obs_mode == 'pixels':                    # the sweep's own time at the end of the soak (library events, 300 steps)
            env.profile_begin(300)
            run(300, t_next)
            torch.cuda.synchronize(dev)
            sp = env.profile_end()
            t_next += 300
        else:
            sp = None
        seg = [b - a for a, b in zip([0.0] + marks[:-1], marks)]
        S_ = args.size
        frame_ = 48 * S_ * S_ if args.raster == 'ray' else 27 * S_ * (S_ + 1)
        soak = {'steps': n_soak, 'seconds': soak_s, 'value': float(N) * n_soak / soak_s, 'unit': 'env-steps/s', 'ms_per_step': soak_s / n_soak * 1e3,
                'episodes_finished_per_step': (int(env.counters[1].item()) - ep0) / max(n_soak + (300 if sp else 0), 1),
                'ms_per_step_by_4096_steps': {'min': min(seg) / 4096 * 1e3, 'max': max(seg) / 4096 * 1e3, 'last': seg[-1] / 4096 * 1e3} if seg else None,
                'tuner_before': tuner0, 'tuner_after': tuner1, 'guard_moves': tuner1['guard_slowdowns'] - tuner0['guard_slowdowns'],
                'sweep_after': ({'avg_launch_ms': sp['ms_render_kernel'], 'median_launch_ms': sp['ms_render_kernel_median'],
                                 'frac': float(N) * (S_ * S_ + frame_) / (sp['ms_render_kernel'] * 1e-3) / 1e9 / HBM_PEAK_GBS} if sp and sp['ms_render_kernel'] > 0 else None),
                'note': 'eager steps of the headline batch for as long as a second run of the CPU oracle ran beside it (one thread of the CPU share left to this loop); the '
                        'host waits for the card every 4 096 steps; episode phases %s' % ('spread out' if (window_desync or args.desync) else 'in step')}

    # side measurements (rank 0's GPU only, short): the same batch in the two cheaper observation modes.
    # They are NOT the headline: pixels_dirty produces the identical frames by repainting <= 2 cells per
    # step (the reference's own render_edit strategy), state-only has no frames at all.  Both are launch-bound: `value` is the per-step
    # Python loop (a policy in the loop), beside it the same steps through cw_step_many (one library call) and as a HIP graph of 64 steps
    # that re-reads an action ring (VecEnv.capture_steps): the card's own pace.
    def side_modes(n_envs, modes, acts):
        out_ = {}
        for mode in modes:
            e2 = CraftingWorldVecEnv(n_envs, size=(args.size, args.size), max_steps=args.max_steps, obs_mode=mode, device=dev, seed=lo)
            e2.reset()
            k2 = 2 * args.max_steps
            for t in range(20):
                e2.step_async(acts[t % rows])
            torch.cuda.synchronize(dev)
            dts = []
            for rep in range(3):                         # three times back to back: box / run variance of the launch-bound modes in one record
                tt = time.perf_counter()
                for t in range(k2):
                    e2.step_async(acts[(20 + rep * k2 + t) % rows])
                torch.cuda.synchronize(dev)
                dts.append(time.perf_counter() - tt)
            dt = dts[0]
            rec = {'value': n_envs * k2 / dt, 'unit': 'env-steps/s', 'ms_per_step': dt / k2 * 1e3, 'steps': k2, 'n_gpus': 1, 'launch': 'eager, one cw_step per Python call',
                   'repeats': {'n': 3, 'value': [n_envs * k2 / x for x in dts], 'us_per_step': [x / k2 * 1e6 for x in dts]}}
            # the step kernel's own time (library events around cw_step_fused_kernel, 200 steps) against the HBM roof by SURVEY 8d's bytes per env-step:
            # A = 48 (state) / 48 + 96 (the reference's dirty-cell repaint).  Neither is bandwidth-bound, and the block says what bounds it instead.
            e2.profile_begin(200)
            for t in range(200):
                e2.step_async(acts[(7 + t) % rows])
            torch.cuda.synchronize(dev)
            pk = e2.profile_end()
            a_bytes = 48.0 + (96.0 if mode == 'pixels_dirty' else 0.0)
            k_ms = pk['ms_step_kernel']
            rec['roofline'] = {
                'bound': 'launch + end-of-kernel write-back' if mode == 'pixels_dirty' else 'launch / latency', 'kernel': 'cw_step_fused_kernel',
                'algorithmic_bytes_per_env_step': a_bytes, 'algorithmic_bytes_per_launch': a_bytes * n_envs, 'avg_launch_ms': k_ms,
                'achieved': a_bytes * n_envs / (k_ms * 1e-3) / 1e9 if k_ms > 0 else None, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': a_bytes * n_envs / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if k_ms > 0 else None,
                'step_frac': a_bytes * n_envs / (dt / k2) / 1e9 / HBM_PEAK_GBS,
                'counters': 'profiles/r05_pmc_dirty.json (rocprofv3 --pmc, one pass per counter group, both modes)',
                'what_bounds_it': ('a wave lives ~6 us of a 13-us launch, one more than in the state-only mode (SQ_WAVE_CYCLES); the rest is dispatch and the end-of-kernel write-back of the L2: '
                                   '163 000 dirty lines against 21 000 in the state-only mode, 129 000 of them partial (32-byte requests with byte masks, '
                                   'TCC_EA0_WRREQ - TCC_EA0_WRREQ_64B) -- the cost follows the number of LINES touched (~3 per move: two pixel rows of two '
                                   'cells), not requests or bytes: the same stores into two lines per env cost a third, whole-line or nontemporal stores '
                                   'cost more (profiles/r05_experiments.txt B)') if mode == 'pixels_dirty' else
                                  ('one kernel launch: 3.6 us at its shortest, 6.2 us on average -- dispatch, one round trip to HBM for 48 bytes per env, '
                                   'the write-back of 21 000 lines; 0.6 TB/s of algorithmic bytes is what that leaves')}
            block = acts[:k2].contiguous()
            kb = int(block.shape[0])                     # (the action pool holds `rows` rows: fewer than k2 when max_steps > 512)
            e2.step_many(block)
            torch.cuda.synchronize(dev)
            tt = time.perf_counter()
            for rep in range(3):
                e2.step_many(block)
            torch.cuda.synchronize(dev)
            dt = (time.perf_counter() - tt) / 3
            rec['step_many'] = {'value': n_envs * kb / dt, 'us_per_step': dt / kb * 1e6, 'launch': 'cw_step_many: %d steps per library call' % kb}
            ring = acts[:64].clone()
            g = e2.capture_steps(ring)
            reps = max(1, k2 // 64)
            for _ in range(2):
                g.replay()
            torch.cuda.synchronize(dev)
            tt = time.perf_counter()
            for rep in range(3 * reps):
                g.replay()
            torch.cuda.synchronize(dev)
            dt = (time.perf_counter() - tt) / (3 * reps * 64)
            rec['graph_replay'] = {'value': n_envs / dt, 'us_per_step': dt * 1e6, 'launch': 'HIP graph of 64 steps reading an action ring (VecEnv.capture_steps)'}
            if mode == 'state':
                e2.rollout(block, record=False)
                torch.cuda.synchronize(dev)
                tt = time.perf_counter()
                e2.rollout(block, record=False)
                torch.cuda.synchronize(dev)
                dt = time.perf_counter() - tt
                rec['rollout'] = {'value': n_envs * kb / dt, 'us_per_step': dt / kb * 1e6, 'launch': 'cw_rollout: %d steps in persistent launches of max_steps steps, a look-ahead refill ahead of each' % kb}
            out_[mode] = rec
            e2.close()
        return out_

    # ... and what finishing episodes cost when a policy SUCCEEDS (round 6): the metric's random policy ends an episode every max_steps steps; a
    # stand-in for a competent one -- 8x8 grids, reward_style='subset', the one task EatBread, a random walker -- ends one every ~140 steps: ~480 of 65 536 envs
    # finish on every step, many of them twice or more between two look-ahead refills of the static period.
    def short_episodes(n_envs, modes):
        out_ = {}
        walk = torch.randint(0, 4, (256, n_envs), device=dev, dtype=torch.uint8, generator=gen)
        for mode in modes:
            e3 = CraftingWorldVecEnv(n_envs, size=(8, 8), max_steps=args.max_steps, obs_mode=mode, device=dev, seed=lo, reward_style='subset',
                                     selected_tasks=['EatBread'], number_of_tasks=1)
            e3.reset()
            for t in range(2 * args.max_steps):
                e3.step_async(walk[t % 256])
            torch.cuda.synchronize(dev)
            c0 = e3._counters_raw.cpu().clone()
            k3 = 2000
            tt = time.perf_counter()
            for t in range(k3):
                e3.step_async(walk[t % 256])
            torch.cuda.synchronize(dev)
            dt = time.perf_counter() - tt
            c1 = e3._counters_raw.cpu()
            fin = float(c1[1] - c0[1])
            out_[mode] = {'value': n_envs * k3 / dt, 'unit': 'env-steps/s', 'us_per_step': dt / k3 * 1e6, 'steps': k3,
                          'episodes_finished_per_step': fin / k3, 'mean_episode_length': n_envs * k3 / max(fin, 1.0),
                          'slow_path_resets_per_step': float(c1[5] - c0[5]) / k3,
                          'workload': '%d envs, 8x8, max_steps=%d, reward_style subset, selected_tasks [EatBread], uniform random MOVES: a stand-in for a policy that '
                                      'succeeds (round 5, one look-ahead record per env and a refill every 64 steps: 21.0 / 26.6 us per step state-only / dirty-cell)'
                                      % (n_envs, args.max_steps)}
            e3.close()
        return out_

    other, config1, short_eps = {}, None, None
    if rank == 0 and world == 1 and not args.no_other_modes and args.obs_mode == 'pixels' and args.raster == 'ray':
        env.close()
        short_eps = short_episodes(N, ('state', 'pixels_dirty'))
        other = side_modes(N, ('pixels_dirty', 'state'), actions)
        # BASELINE configs[1]: 4 096 envs, default grid, state-only obs, random actions
        config1 = side_modes(4096, ('state',), actions[:, :4096].contiguous())['state']
        config1['workload'] = 'BASELINE configs[1]: 4096 envs, %dx%d, state-only obs, uniform random actions' % (args.size, args.size)

    if rank == 0:
        total_steps = float(N) * world * K
        value = total_steps / elapsed
        S = args.size
        frame = 48 * S * S if args.raster == 'ray' else 27 * S * (S + 1)
        if args.obs_mode == 'pixels':
            # A_pix = 48 + S*S (grid read) + frame write; the sweep's share is S*S + frame bytes per env, and one sweep
            # paints N envs.  (The INIT_OBS / desired_goal frames of envs that finished -- 2 x resets_in_prof frames in the profiled region --
            # are painted inside cw_step_fused_kernel, BEFORE the sweep and outside the bracketed kernel: not counted here.)
            alg_bytes = plain_alg_bytes = float(N) * (S * S + frame)
            dominant, ms = render_kernel, prof['ms_render_kernel']
        else:
            alg_bytes = plain_alg_bytes = float(N) * 48.0
            # state-only / dirty-cell: the whole auto-reset step is one launch (step + inline resets), latency-bound
            dominant, ms = 'cw_step_fused_kernel', prof['ms_step_kernel']
        achieved = alg_bytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        step_alg_bytes = float(N) * ((48.0 + S * S + frame) if args.obs_mode == 'pixels' else (48.0 + 96.0) if args.obs_mode == 'pixels_dirty' else 48.0)
        # HBM bytes per launch of the dominant kernel from the PMC passes (tools/profile_pmc.sh: separate
        # WRITE_SIZE / FETCH_SIZE runs, calibrated; the newest committed summary is quoted, null otherwise)
        traffic = traffic_source = None
        import glob
        mine = {'envs_per_gpu': N, 'size': S, 'obs_mode': args.obs_mode, 'raster': args.raster,
                'episode_phases': 'spread out (--desync)' if args.desync else 'synchronized start',
                'task_lists': 'eight ordered menus, env i uses menu i mod 8' if args.mixed_menus else 'one (all nine tasks)'}
        for f in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_traffic*.json')), reverse=True):      # the newest committed summary of THIS shape
            try:
                pj = json.load(open(f))
            except Exception:  # noqa: BLE001
                continue
            if pj.get('shape') == mine and pj.get('hbm_bytes_per_launch'):
                traffic = pj['hbm_bytes_per_launch'] * pj.get('launches_per_sweep', 1)      # (a sweep of several chunk launches: the bracketed region holds them all)
                traffic_source = ('not measured in this run: quoted from %s (separate rocprofv3 --pmc passes of the same shape, '
                                  'tools/profile_pmc.sh)' % os.path.relpath(f, ROOT))
                break
        # the practical ceiling beside the spec peak: a plain device fill of the same number of bytes,
        # timed on this box right now (torch's vectorised fill kernel, median of 9)
        fill_gbs = None
        if world == 1 and args.obs_mode == 'pixels':
            probe = torch.empty(int(alg_bytes), dtype=torch.uint8, device=dev)
            ts = []
            for i in range(11):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); probe.fill_(i); e1.record(); e1.synchronize()
                ts.append(e0.elapsed_time(e1))
            ts = sorted(ts[2:])
            fill_gbs = alg_bytes / (ts[len(ts) // 2] * 1e-3) / 1e9
            del probe
        out = {
            'metric': 'env-steps/sec at 65536 envs, 1/2/4/8 MI355X; bit-exact vs CPU ref',
            'value': value, 'unit': 'env-steps/s', 'n_gpus': world, 'steps': K, 'warmup': W,
            'ms_per_step': elapsed / K * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'u8', 'data': 'synthetic',
            # the same K steps by the wall clock around the whole bracket (opening barrier ... closing barrier + synchronize), max over ranks
            'value_wall': total_steps / elapsed_wall, 'ms_per_step_wall': elapsed_wall / K * 1e3,
            'timing': 'value / ms_per_step: the slowest rank\'s DEVICE time for the K steps -- two events on the launch stream inside the barrier + synchronize '
                      'bracket, the first right behind the opening barrier, the second behind the last step; value_wall: the wall clock around the bracket, '
                      'which also holds the closing barrier (per_rank[].barrier_ms) and the host\'s wake-up',
            'config': {'workload': 'BASELINE configs[2]: %d envs/GPU, %dx%d grid, %s obs, auto-reset, uniform random '
                                   'actions, max_steps=%d' % (N, S, S, {'pixels': 'full-frame 4x4-cell uint8 pixel',
                                                                        'pixels_dirty': 'dirty-cell-repaint pixel',
                                                                        'state': 'state-only'}[args.obs_mode], args.max_steps),
                       'envs_per_gpu': N, 'size': S, 'max_steps': args.max_steps, 'obs_mode': args.obs_mode, 'raster': args.raster,
                       'sharding': 'contiguous env ranges per rank, no data-path collective',
                       'task_lists': 'eight ordered menus, env i uses menu i mod 8' if args.mixed_menus else 'one (all nine tasks)',
                       'launch': launch_desc, 'episode_phases': 'spread out (--desync)' if args.desync else 'synchronized start',
                       'timed_region': ('placed between two of the steps on which every env times out at once (K < max_steps: %d untimed steps '
                                        'before the W warm-up steps instead of %d); metric_window holds them in their true proportion'
                                        % (prewarm_steps, max(args.prewarm_steps, 0))) if prewarm_steps != max(args.prewarm_steps, 0) else
                                       'the first K steps after the warm-up, wherever the all-env time-out steps fall'},
            # the figures to quote: the 2*max_steps window with both all-env time-out steps inside, and the same with the episode phases spread out
            'quote': {'metric_window': window['value'] if window else None, 'metric_window_desync': window_desync['value'] if window_desync else None},
            'roofline': {'bound': 'hbm', 'kernel': dominant,
                         # (a rocprofv3 trace lists the sweep by raster and frames per job: cw_render_pieces_kernel<raster, 2> for frames of 4 KiB and more)
                         'kernel_in_trace': (('%s<%d, %d>' % (dominant, 1 if args.raster == 'alt' else 0, frames_per_job(frame))) if dominant.startswith('cw_render_pieces') else
                                             ('%s<%d>' % (dominant, 4 if frames_per_job(frame) <= 4 else 8)) if dominant.startswith('cw_render_gather') else dominant),
                         'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic, 'traffic_source': traffic_source,
                         # the WHOLE step against the same roof: A x N / ms_per_step (A = SURVEY 8d's bytes per env-step: 48 + S*S + frame in the
                         # full-frame mode) for `value` and for the two windows to quote
                         'step_frac': step_alg_bytes / (elapsed / K) / 1e9 / HBM_PEAK_GBS,
                         'step_frac_metric_window': step_alg_bytes / (window['ms_per_step'] * 1e-3) / 1e9 / HBM_PEAK_GBS if window else None,
                         'step_frac_metric_window_desync': (step_alg_bytes / (window_desync['ms_per_step'] * 1e-3) / 1e9 / HBM_PEAK_GBS
                                                            if window_desync else None),
                         'step_algorithmic_bytes': step_alg_bytes,
                         'fill_same_bytes_GBs': fill_gbs,   # plain fill of the same size on this box, for orientation
                         'algorithmic_bytes_per_launch': alg_bytes, 'avg_launch_ms': ms,
                         'launch_ms_min_max': [prof['ms_render_kernel_min'], prof['ms_render_kernel_max']],
                         # the average above includes the launches on which every env is reset inside the kernel (2 of 600 with
                         # synchronized phases, ~4x a plain launch, latency-bound): the median launch and its plain byte count beside it
                         'median_launch_ms': prof['ms_render_kernel_median'] or None,
                         'frac_at_median_launch': (plain_alg_bytes / (prof['ms_render_kernel_median'] * 1e-3) / 1e9 / HBM_PEAK_GBS
                                                   if args.obs_mode == 'pixels' and prof['ms_render_kernel_median'] > 0 else None),
                         'events': 'hipEventRecord on the launch stream around every kernel, %d launches' % prof['steps']},
            # full-pixel mode brackets only the dominant kernel, the sweep (each event record costs a pipeline bubble)
            'kernels_ms': {'step': prof['ms_step_kernel'] or None, 'reset': prof['ms_reset_kernel'] or None,
                           'render': prof['ms_render_kernel'] or None, 'ms_per_step_with_events': elapsed_prof / K * 1e3},
            'tuner': tuner,
            'episodes_finished': episodes, 'prewarm_steps': prewarm_steps,
            'warmup_total': prewarm_steps + W,           # untimed steps before the timed region: `warmup` is the contract's W
            'per_rank_ms_per_step': [x / K * 1e3 for x in per_rank_s],
            'per_rank_ms_per_step_wall': [x / K * 1e3 for x in per_rank_wall_s],
            'per_rank_tuner': [r_['tuner'] for r_ in per_rank],
            'per_rank_roofline_frac': [r_['roofline_frac'] for r_ in per_rank],
            'per_rank': per_rank,
            'repeats': {'n': len(repeats_s), 'ms_per_step': [x / K * 1e3 for x in repeats_s],
                        'value': [total_steps / x for x in repeats_s], 'value_wall': [total_steps / x for x in repeats_wall_s], 'value_min': total_steps / max(repeats_s),
                        'value_median': total_steps / sorted(repeats_s)[len(repeats_s) // 2],
                        'note': 'the K-step region three times back to back; `value` is the first'},
            'metric_window': window,
            'metric_window_desync': window_desync,
            'dist_backend': backend_used,
            'other_obs_modes_1gpu': other,
            'config1_state_4096': config1,
            'short_episodes_1gpu': short_eps,
            'policy_in_loop': policy_blocks,
        }
        if cpu_result is not None:                       # rank 0 at N=1 only (task contract); measured above, beside the GPU soak
            out['cpu_baseline'] = cpu_result
            out['soak_beside_cpu_baseline'] = soak
        if not args.no_single_env and world == 1:
            env.close()                                  # (idempotent)
            out['single_env'] = single_env_latency(dev)
    env.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        sys.stdout.flush()
        json_out.write(json.dumps(out) + '\n')         # the last line of rank 0's stdout
        json_out.flush()


if __name__ == '__main__':
    main()
