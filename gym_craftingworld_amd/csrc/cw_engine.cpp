// cw_engine.cpp -- host side of the C ABI in include/craftingworld.h: engine lifetime, device
// buffers, MT19937 state conversion, dense<->slot state conversion, kernel launches.
// No CPU fallback exists: every compute entry point enqueues HIP kernels (cw_kernels.hip).
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <algorithm>
#include <vector>

#include "../../include/craftingworld.h"
#include "cw_layout.h"
#include "cw_host.h"

extern "C" {
hipError_t cwk_launch_step(const CwParams *P, const CwTuning *T, const void *actions, int act_dtype, int obs_mode, int auto_reset, hipStream_t st,
                           hipEvent_t *ev);
hipError_t cwk_launch_refill(const CwParams *P, const CwTuning *T, int all_envs, hipStream_t st);
hipError_t cwk_launch_reset_all(const CwParams *P, const CwTuning *T, int obs_mode, hipStream_t st);
hipError_t cwk_launch_pool(const CwParams *P, const CwTuning *T, hipStream_t st);
hipError_t cwk_launch_seed(const CwParams *P, const uint32_t *seeds_dev, hipStream_t st);
hipError_t cwk_launch_resident(const CwParams *P, CwResident *R, uint32_t seq0, int paint_dirty, unsigned long long idle_ticks,
                               unsigned long long life_ticks, hipStream_t st);
hipError_t cwk_launch_render_onehot(const CwParams *P, const uint8_t *onehot, int n_states, uint16_t *out, hipStream_t st);
hipError_t cwk_launch_render_restore(const CwParams *P, const CwTuning *T, hipStream_t st);
hipError_t cwk_launch_rollout(const CwParams *P, const uint8_t *actions, int T, int32_t *rewards, uint8_t *dones, hipStream_t st);
hipError_t cwk_launch_render_ext(const CwParams *P, const CwTuning *T, uint8_t *out, hipStream_t st);
hipError_t cwk_launch_sweep_calib(const CwParams *P, const CwTuning *T, hipStream_t st);
void cwk_sweep_shape(const CwParams *P, const CwTuning *T, int *n_chunks, int *waves, int *jobs_per_wave);
hipError_t cwk_launch_idle(hipStream_t st);
hipError_t cwk_launch_export(const CwParams *P, const CwTuning *T, uint8_t *out, int onehot, int which, hipStream_t st);
}

#include <time.h>

static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                           \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess)                                                                   \
            return fail(CW_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = (hipSetDevice(dev) == hipSuccess);
    }
    ~DeviceGuard()
    {
        int cur = -1;
        if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
    }
};

struct cw_engine {
    int device = 0;
    int obs_mode = 0;
    int auto_reset = 1;
    bool has_reset = false;
    CwParams P{};
    CwTuning tune{};
    std::vector<void *> allocs;
    std::vector<void *> host_allocs;   // hipHostMalloc'ed (cw_config.host_outputs)
    int32_t *host_actions = nullptr;
    std::vector<CwMenuDev> menus;
    uint32_t *seed_scratch = nullptr;  // [N] device: seeds of cw_seed_int
    int n = 0, S = 0, ncell = 0, K = 0;
    // resident stepper of the single-env loop (cw_step_resident; cw_kernels.hip: cw_resident_kernel)
    CwResident *res = nullptr;         // pinned coherent host memory, or null (engine not eligible)
    hipStream_t res_stream = nullptr;
    bool res_running = false;          // a cw_resident_kernel may be on the card
    uint32_t res_seq = 0;              // last sequence number rung
    hipEvent_t last_work = nullptr;    // recorded after the last cw_reset / cw_step / cw_rollout on the caller's stream: a resident kernel starts only after that
    bool last_work_set = false;        //   work (an event of the engine's own: the caller may destroy its stream any time after cw_synchronize)
    unsigned long long res_ticks_per_us = 100;   // the device's constant clock (wall_clock64), hipDeviceAttributeWallClockRate
    std::vector<hipEvent_t> prof_ev;   // 6 per recorded step
    // look-ahead (cw_layout.h): the refill kernel is launched every e->la_period steps, ahead of the step, on the step's stream
    bool la_refill_all = false;        // the next refill covers every env without a record (after cw_reset / a re-seed / a checkpoint load)
    unsigned la_steps = 0;
    unsigned la_period = 16;           // the CURRENT refill period: la_period_max, or shorter while envs finish twice between two refills (la_adapt)
    unsigned la_period_max = 16;       // la_period_for(max_steps), or CW_TUNE_LA_PERIOD
    bool la_adaptive = true;
    int rollout_segment = -1;          // CW_TUNE_ROLLOUT_SEGMENT (read at cw_create like every tuning variable): steps per persistent launch of cw_rollout, 0: one launch, -1: max_steps
    unsigned long long *la_feedback = nullptr;     // pinned: counters[5] as the last refill kernel saw it
    unsigned long long la_slow_seen = 0;
    int32_t la_quiet = 0;              // refills in a row with (nearly) no slow-path reset
    bool in_step_many = false;         // (cw_step_many decides about the refill of a captured sequence itself)
    bool capturing_now = false;        // ... and asks once whether its stream is capturing, for all of its steps
    // the sweep's clock (calibrate_sweep) and its guard (sweep_guard_tick)
    int sweep_waves = 1024;            // waves of a sweep's launch, jobs (4-KiB pieces) per wave over all of its launches
    double sweep_jobs = 0, sweep_rate = 0, sweep_beside_ms = 0;      // (sweep_rate: cw_create's choice; the live one is guard.rate)
    bool guard_on = false;
    // the guard's samples in flight: the host runs up to ~1 000 steps ahead of the card, so a sample recorded now is read a dozen samples later
    enum { GUARD_RING = 32 };
    struct GuardSample { hipEvent_t ev[6]; int period16; } guard_ring[GUARD_RING] = {};
    unsigned guard_head = 0, guard_tail = 0;      // next slot to record into / oldest slot not read yet
    unsigned guard_step = 0;
    cwh_guard guard{};                  // its decisions (cw_host.cpp: cwh_guard_step): rate, best rate known, trials, back-off
    int prof_cap = 0, prof_n = 0;
    // the engine's OWN work: the streams it was handed since its last wait (at most 4 are remembered) and a private stream for the synchronous
    // entry points' copies and kernels (cw_seed_*, cw_get_mt, cw_get/set_state, checkpoints): none of them waits for anybody else's work
    hipStream_t aux = nullptr;
    hipStream_t work[4] = {nullptr, nullptr, nullptr, nullptr};
    int n_work = 0;
    bool work_overflow = false;        // more than 4 distinct streams since the last wait: a device-wide wait is the only safe one
    bool captured = false;             // STICKY: a step / rollout of this engine has been captured into a HIP graph.  Replays run on whatever stream the caller
                                       // launches the graph on, which the engine never sees: from then on every synchronous entry point waits for the DEVICE
};
// is `st` recording a graph right now?  (an error counts as no: the launch that follows reports it)
static inline bool stream_capturing(hipStream_t st)
{
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    return hipStreamIsCapturing(st, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone;
}
// every entry point that ENQUEUES on a caller's stream notes it (NOT a capturing stream: nothing runs on it, and it may be gone when the graph is replayed) ...
static inline void note_work(cw_engine *e, hipStream_t st)
{
    for (int i = 0; i < e->n_work; i++) if (e->work[i] == st) return;
    if (e->n_work < 4) e->work[e->n_work++] = st; else e->work_overflow = true;
}
// ... and every synchronous one first waits for exactly that: the engine's own work, not the card's (two engines on one device, a learner beside
// them: MultiDeviceVecEnv).  A stream the caller destroyed meanwhile (its work has then been waited for or abandoned by the caller) fails the
// wait: then, and only then, the whole device is waited for.
static hipError_t quiesce(cw_engine *e)
{
    bool all = e->work_overflow || e->captured;
    for (int i = 0; i < e->n_work && !all; i++)
        if (hipStreamSynchronize(e->work[i]) != hipSuccess) { (void)hipGetLastError(); all = true; }
    e->n_work = 0;
    e->work_overflow = false;
    if (all) return hipDeviceSynchronize();
    return e->aux ? hipStreamSynchronize(e->aux) : hipSuccess;
}
// The synchronous entry points copy with hipMemcpyAsync on e->aux into / out of vectors and stack buffers of their own: whatever way such a function
// is left -- an early return on an error included -- nothing may still be in flight against storage that dies with its frame.
struct AuxDrain {
    cw_engine *e;
    explicit AuxDrain(cw_engine *e_) : e(e_) {}
    ~AuxDrain() { if (e && e->aux) (void)hipStreamSynchronize(e->aux); }
};
static inline hipError_t aux_copy(cw_engine *e, void *dst, const void *src, size_t bytes, hipMemcpyKind kind)
{
    return bytes ? hipMemcpyAsync(dst, src, bytes, kind, e->aux) : hipSuccess;
}
// look-ahead refill: every la_period-th step.  A refill costs one reset's latency (~15 us) whatever the list holds, so rarely is cheap -- but an env
// that finishes twice between two refills is reset the slow way: a quarter of the episode length, 8..64 steps
static int la_period_for(int max_steps) { const int p = max_steps / 4; return p < 8 ? 8 : p > 64 ? 64 : p; }
// THE REFILL PERIOD FOLLOWS THE EPISODES (round 6).  la_period_for() assumes episodes of about max_steps steps -- a random policy's.  A policy that SUCCEEDS ends
// its episodes much earlier: an env then finishes two or three times between two refills, finds no record the second time and is reset the slow way by its
// whole wave -- ~12 us on the step kernel's critical path, and the kernel ends with its slowest wave (a walker on 8x8 grids under max_steps 300, episodes of
// ~140 steps: 98 slow resets per step at the static period of 64, state-only step 21 us instead of 6; profiles/r06_experiments.txt D).  Every refill kernel
// leaves the count of slow resets in a pinned word; the host reads it when it enqueues the next refill -- stale by a period or more, never waited for -- and
// halves the period (8 at least) while a period brings more slow resets than an eighth of its steps (what a refill launch costs), doubles it back after eight
// quiet refills in a row.  Results do not depend on the period; a captured graph keeps the period it was captured with.
static void la_adapt(cw_engine *e)
{
    const unsigned long long slow = __atomic_load_n(e->la_feedback, __ATOMIC_RELAXED);
    const unsigned long long delta = slow >= e->la_slow_seen ? slow - e->la_slow_seen : 0;      // (a loaded checkpoint brings its own counters)
    e->la_slow_seen = slow;
    e->la_period = (unsigned)cwh_la_adapt((int32_t)e->la_period, (int32_t)e->la_period_max, delta, &e->la_quiet);      // (cw_host.cpp: the rule, tested on the CPU)
}

// ------------------------------------------------------------------------------ resident stepper (host side)
// Every entry point that reads or writes the engine's state first makes sure no resident kernel holds it in registers.
static int resident_park(cw_engine *e)
{
    if (!e->res_running) return CW_OK;
    const unsigned long long low = __atomic_load_n(&e->res->bell, __ATOMIC_RELAXED) & 0xFFFFFFFFull;
    __atomic_store_n(&e->res->bell, low | (1ull << 32), __ATOMIC_RELEASE);
    HIP_TRY(hipStreamSynchronize(e->res_stream));        // the kernel leaves within one poll of seeing the flag
    e->res_running = false;
    __atomic_store_n(&e->res->bell, low, __ATOMIC_RELEASE);
    __atomic_store_n(&e->res->exited, 0u, __ATOMIC_RELEASE);
    return CW_OK;
}
#define PARK(e) do { if ((e)->res_running) { const int _rc = resident_park(e); if (_rc != CW_OK) return _rc; } } while (0)

// (MT19937 state conversion, the DLPack producer, the dense view of a slot record, the checkpoint sections' sizes and the guard's decisions: cw_host.cpp)

// ------------------------------------------------------------------------------ helpers
template <typename T>
static int dev_alloc(cw_engine *e, T **out, size_t count)
{
    void *p = nullptr;
    size_t bytes = count * sizeof(T);
    if (bytes == 0) bytes = 16;
    HIP_TRY(hipMalloc(&p, bytes));
    HIP_TRY(hipMemset(p, 0, bytes));
    e->allocs.push_back(p);
    *out = (T *)p;
    return CW_OK;
}

// pinned host memory mapped into the device's address space: kernels store into it directly (over PCIe), the host
// reads it after a stream sync -- the single-env loop's outputs (cw_config.host_outputs)
template <typename T>
static int host_alloc(cw_engine *e, T **out, size_t count)
{
    void *p = nullptr, *d = nullptr;
    size_t bytes = count * sizeof(T);
    if (bytes == 0) bytes = 16;
    HIP_TRY(hipHostMalloc(&p, bytes, hipHostMallocCoherent));       // (fine-grained: a resident kernel's stores must reach the host while it runs)
    e->host_allocs.push_back(p);
    memset(p, 0, bytes);
    HIP_TRY(hipHostGetDevicePointer(&d, p, 0));
    if (d != p) return fail(CW_ERR_HIP, "cw_create: mapped host memory has a different device address (no unified addressing)");
    *out = (T *)p;
    return CW_OK;
}

static void prof_free(cw_engine *e)
{
    for (hipEvent_t ev : e->prof_ev) (void)hipEventDestroy(ev);
    e->prof_ev.clear();
    e->prof_cap = e->prof_n = 0;
}

// launch time of the per-step render as currently configured (launches queued back to back, one wait: see calibrate_render_pace): the median and the
// 90th percentile of 20 launches -- the write path has a slower regime that a configuration may enter launch by launch (profiles/r03_pieces.txt N-P), and
// a median does not show a configuration that does so one time in three
static int timed_render_stats(cw_engine *e, double *median, double *p90, double *mean = nullptr)
{
    enum { LAUNCHES = 24, SKIP = 4 };
    hipEvent_t evs[2 * LAUNCHES] = {};
    bool ok = true;
    for (hipEvent_t &ev : evs) ok = ok && hipEventCreate(&ev) == hipSuccess;
    for (int rep = 0; rep < LAUNCHES && ok; rep++)
        ok = hipEventRecord(evs[2 * rep], e->aux) == hipSuccess && cwk_launch_sweep_calib(&e->P, &e->tune, e->aux) == hipSuccess &&
             hipEventRecord(evs[2 * rep + 1], e->aux) == hipSuccess && cwk_launch_idle(e->aux) == hipSuccess;
    ok = ok && hipStreamSynchronize(e->aux) == hipSuccess;
    float ms[LAUNCHES];
    for (int rep = 0; rep < LAUNCHES && ok; rep++) ok = hipEventElapsedTime(&ms[rep], evs[2 * rep], evs[2 * rep + 1]) == hipSuccess;
    for (hipEvent_t &ev : evs) if (ev) (void)hipEventDestroy(ev);
    if (!ok) return fail(CW_ERR_HIP, "cw_create: render calibration failed");
    std::sort(ms + SKIP, ms + LAUNCHES);
    const int n = LAUNCHES - SKIP;
    *median = ms[SKIP + n / 2];
    *p90 = ms[SKIP + (9 * n) / 10 - 1];
    if (mean) { double acc = 0; for (int i = SKIP; i < LAUNCHES; i++) acc += ms[i]; *mean = acc / n; }
    return CW_OK;
}

// The sweep's CLOCK (cw_kernels.hip: render_pieces): the period between two jobs of a wave, i.e. the RATE at which a launch writes --
// waves x 4 KiB per period.  In a row of back-to-back launches the memory system of an MI355X takes up to 7.7 TB/s of such a stream (65 536 envs,
// 21x21, 1 024 waves, 545 ns: 0.1988 ms per launch in step, 0.2004 with ~220 envs finishing on every step, 0.89 / 0.88 of the 8 TB/s peak) IF the
// launch's first ~40 us run a notch slower (cw_render_pieces_kernel: the head), and falls into its slower, saturated regime just beyond (525 ns:
// 0.20-0.24 whatever the head) or, box by box, already there (one launch in four 10-15 % late) -- profiles/r04_clock.txt K-M.  So cw_create tries 7.7,
// 7.4, 7.2, 7.0, 6.8, 6.6 TB/s and the unclocked sweep on the engine's own batch (20 launches each, an idle kernel between them like a step's)
// and takes the best 90th percentile -- a rate that is late one time in four loses to the next one -- and a GUARD keeps watching in cw_step
// (sweep_guard_tick): a sweep that does not keep its schedule any more is slowed down a notch.
// CW_TUNE_PERIOD_NS forces a period (0: unclocked, every wave as fast as it can).
static double sweep_period_ns(const cw_engine *e, double tb_per_s) { return (double)e->sweep_waves * 4096.0 / (tb_per_s * 1e12) * 1e9; }
// the clock's three periods from one rate (cwh_sweep_periods): a launch's first CWH_HEAD_JOBS jobs run CW_HEAD_NOTCH slower, after a step on which envs
// finished CW_BUSY_NOTCH slower (cw_render_pieces_kernel)
static double CW_HEAD_NOTCH = 0.4, CW_BUSY_NOTCH = 0.75;      // (TB/s; CW_TUNE_HEAD_NOTCH / CW_TUNE_BUSY_NOTCH for experiments: profiles/r05_experiments.txt J)
static void set_sweep_rate(cw_engine *e, double tb_per_s)
{
    cwh_sweep_periods(tb_per_s, e->sweep_waves, CW_HEAD_NOTCH, CW_BUSY_NOTCH, &e->tune.period16, &e->tune.period16_head, &e->tune.period16_busy);
}

static int calibrate_sweep(cw_engine *e)
{
    CwTuning &tn = e->tune;
    int n_chunks = 1, jobs_per_wave = 1;
    cwk_sweep_shape(&e->P, &tn, &n_chunks, &e->sweep_waves, &jobs_per_wave);
    e->sweep_jobs = (double)n_chunks * jobs_per_wave;
    if (const char *v = getenv("CW_TUNE_HEAD_NOTCH")) CW_HEAD_NOTCH = atof(v);
    if (const char *v = getenv("CW_TUNE_BUSY_NOTCH")) CW_BUSY_NOTCH = atof(v);
    if (const char *per = getenv("CW_TUNE_PERIOD_NS")) {                         // (the heads keep their distance: a forced 545 ns is 7.7 TB/s with 7.3 / 6.95 heads)
        const double ns = atof(per);
        set_sweep_rate(e, ns > 0 ? (double)e->sweep_waves * 4096.0 / (ns * 1e-9) * 1e-12 : 0.0);
        if (ns > 0) tn.period16 = (int)(ns * 1.6 + 0.5);
        return CW_OK;
    }
    set_sweep_rate(e, 7.0);
    // (small batches are launch-bound: nothing to check; host-mapped frames are PCIe-bound: unclocked)
    if (e->host_actions) { set_sweep_rate(e, 0); return CW_OK; }
    if (e->obs_mode != CW_OBS_PIXELS_FULL || (double)e->n * e->P.frame_bytes < (double)(64ll << 20)) return CW_OK;
    // the candidates, the fastest first; unclocked last (a sweep whose jobs take longer than any useful period -- several small frames per piece --
    // paces itself: the clock then only costs its reads).  The one with the best 90th-percentile launch: in its saturated regime the memory
    // system is slower AND erratic (7.5 TB/s: 0.214-0.235 ms launch by launch where 7.0 reads 0.2112 +- 0.0005), so the slow launches tell.
    static const double rates[] = {7.7, 7.4, 7.2, 7.0, 6.8, 6.6, 0.0};
    char log[1024] = "";
    size_t len = 0;
    int rc = CW_OK;
    double best_p90 = 0, med = 0, p90 = 0, mean = 0;
    rc = timed_render_stats(e, &med, &p90);           // (a card that idled through set-up runs its first launches a few per cent slower: not counted)
    // what a sweep costs beside its jobs (a launch's ramp and tail, the events around it), at a rate the memory system keeps up with easily: the guard's yardstick
    set_sweep_rate(e, 6.4);
    if (rc == CW_OK) rc = timed_render_stats(e, &med, &p90);
    e->sweep_beside_ms = med - e->sweep_jobs * sweep_period_ns(e, 6.4) * 1e-6;
    const char *forced_rate = getenv("CW_TUNE_RATE_TBS");                      // (the starting rate, the guard stays on: the test that it slows a saturated sweep down)
    if (forced_rate && atof(forced_rate) > 0) { e->sweep_rate = atof(forced_rate); best_p90 = 1e-9; }
    for (size_t i = 0; i < sizeof(rates) / sizeof(rates[0]) && rc == CW_OK && !forced_rate; i++) {
        set_sweep_rate(e, rates[i]);
        rc = timed_render_stats(e, &med, &p90, &mean);
        if (len < sizeof(log) - 56) len += (size_t)snprintf(log + len, sizeof(log) - len, " %.1f TB/s (%.0f ns): %.4f/%.4f/%.4f |", rates[i], tn.period16 / 1.6, med, p90, mean);
        if (rc == CW_OK && (best_p90 == 0 || p90 < 0.995 * best_p90)) { best_p90 = p90; e->sweep_rate = rates[i]; }
    }
    set_sweep_rate(e, e->sweep_rate);
    cwh_guard_init(&e->guard, e->sweep_rate);
    e->guard_on = rc == CW_OK && e->sweep_rate > 0 && e->auto_reset && !(getenv("CW_TUNE_GUARD") && atoi(getenv("CW_TUNE_GUARD")) == 0);
    if (e->guard_on)
        for (auto &smp : e->guard_ring)
            for (hipEvent_t &ev : smp.ev)
                if (hipEventCreate(&ev) != hipSuccess) e->guard_on = false;
    if (getenv("CW_TUNE_VERBOSE"))
        fprintf(stderr, "[craftingworld] sweep clock, ms per sweep (median/90th percentile/mean of 20; 0.0 TB/s = unclocked):%s -> %s%.0f ns\n", log,
                tn.period16 ? "" : "unclocked, ", tn.period16 / 1.6);
    return rc;
}

// The GUARD of the sweep's clock, the plumbing: every CW_GUARD_EVERY-th step's sweep is bracketed by two events on the caller's stream; when they have
// completed (read at the next sampled step, however far the host runs ahead) the sweep's time and its schedule go to cwh_guard_step (cw_host.cpp: the
// decisions -- slowdowns, trials one notch up, back-off -- as a pure state machine, tested on synthetic traces in tests/test_host_logic.py), and a
// changed rate is programmed into the clock.  -> the event array for this step's launch, or null.
enum { CW_GUARD_EVERY = 64 };
static hipEvent_t *sweep_guard_tick(cw_engine *e, hipStream_t st)
{
    if (++e->guard_step % CW_GUARD_EVERY) return nullptr;
    const bool verbose = getenv("CW_TUNE_VERBOSE") != nullptr;
    // the samples whose sweeps have run by now, oldest first (the host may be a thousand steps ahead of the card: a sample is read long after it was recorded)
    while (e->guard_tail != e->guard_head && hipEventQuery(e->guard_ring[e->guard_tail % cw_engine::GUARD_RING].ev[5]) == hipSuccess) {
        cw_engine::GuardSample &smp = e->guard_ring[e->guard_tail++ % cw_engine::GUARD_RING];
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, smp.ev[4], smp.ev[5]) != hipSuccess || ms <= 0.f || smp.period16 != e->tune.period16) continue;      // (a sample of another rate says nothing)
        const double scheduled = cwh_guard_scheduled_ms(e->sweep_jobs, e->tune.period16, e->tune.period16_busy, e->sweep_beside_ms);
        const double rate_before = e->guard.rate, prev_mean = e->guard.prev_mean;
        const int recovering = e->guard.recovering;
        const int action = cwh_guard_step(&e->guard, ms, scheduled);
        if (e->guard.rate != rate_before) set_sweep_rate(e, e->guard.rate);
        if (verbose && action != CWH_GUARD_NONE) {
            static const char *what[] = {"", "three samples in a row late: a notch down", "trying a notch up", "the trial pays: kept", "the trial does not pay: undone"};
            fprintf(stderr, "[craftingworld] sweep clock: %s%s -- %.4f ms against %.4f scheduled (mean at the rate left %.4f): %.1f -> %.1f TB/s (%.0f ns)\n", what[action],
                    recovering ? " (on the way back to the best rate known)" : "", ms, scheduled, prev_mean, rate_before, e->guard.rate, e->tune.period16 / 1.6);
        }
    }
    if (e->guard_head - e->guard_tail >= (unsigned)cw_engine::GUARD_RING) return nullptr;      // (every slot in flight: this step goes unsampled)
    cw_engine::GuardSample &slot = e->guard_ring[e->guard_head++ % cw_engine::GUARD_RING];
    slot.period16 = e->tune.period16;
    return slot.ev;
}

extern "C" {

const char *cw_last_error(void) { return g_err; }
int cw_abi_version(void) { return CW_ABI_VERSION; }
int cw_num_envs(const cw_engine *e) { return e ? e->n : 0; }

int cw_create(const cw_config *cfg, int device, cw_engine **out)
{
    if (!cfg || !out) return fail(CW_ERR_INVALID, "cw_create: null argument");
    *out = nullptr;
    if (cfg->abi_version != CW_ABI_VERSION)
        return fail(CW_ERR_INVALID, "cw_create: abi_version %d != %d", cfg->abi_version, CW_ABI_VERSION);
    if (cfg->num_envs < 1) return fail(CW_ERR_INVALID, "cw_create: num_envs must be >= 1");
    if (cfg->size < 4 || cfg->size > 255)
        return fail(CW_ERR_INVALID, "cw_create: size must be in 4..255 (square grids only; non-square is a reference defect)");
    if (cfg->max_steps < 1 || cfg->max_steps > 65535) return fail(CW_ERR_INVALID, "cw_create: max_steps must be in 1..65535");
    if (cfg->n_task_list < 9 || cfg->n_task_list > CW_MAX_TASKS)
        return fail(CW_ERR_INVALID, "cw_create: len(task_list) must be in 9..%d", CW_MAX_TASKS);
    if (cfg->fixed_init_state < 0 || cfg->fixed_init_state > 64)
        return fail(CW_ERR_INVALID, "cw_create: fixed_init_state must be in 0..64");
    if (cfg->obs_mode < CW_OBS_STATE || cfg->obs_mode > CW_OBS_PIXELS_DIRTY) return fail(CW_ERR_INVALID, "cw_create: bad obs_mode");
    if (cfg->raster != CW_RASTER_RAY && cfg->raster != CW_RASTER_ALT) return fail(CW_ERR_INVALID, "cw_create: bad raster");
    if (cfg->n_menus < 1 || cfg->n_menus > CW_MAX_MENUS || !cfg->menus) return fail(CW_ERR_INVALID, "cw_create: n_menus must be in 1..%d", CW_MAX_MENUS);
    if ((double)cfg->num_envs * 48.0 * cfg->size * cfg->size > 1.0e12)
        return fail(CW_ERR_INVALID, "cw_create: num_envs x frame size exceeds 1 TB");

    std::vector<CwMenuDev> menus(cfg->n_menus);
    for (int m = 0; m < cfg->n_menus; m++) {
        const cw_task_menu &src = cfg->menus[m];
        if (src.n_selected < 1 || src.n_selected > CW_MAX_TASKS)
            return fail(CW_ERR_INVALID, "cw_create: menu %d: len(selected_tasks) must be in 1..%d", m, CW_MAX_TASKS);
        if (src.number_of_tasks < 1) return fail(CW_ERR_INVALID, "cw_create: menu %d: number_of_tasks must be >= 1", m);
        CwMenuDev d{};
        d.n_selected = src.n_selected;
        d.number_of_tasks = src.number_of_tasks > src.n_selected ? src.n_selected : src.number_of_tasks;  // ray.py:80-81
        d.stacking = src.stacking ? 1 : 0;
        d.reward_subset = src.reward_subset ? 1 : 0;
        for (int i = 0; i < src.n_selected; i++) {
            if (src.selected_bits[i] < 0 || src.selected_bits[i] >= cfg->n_task_list)
                return fail(CW_ERR_INVALID, "cw_create: menu %d: selected task %d is not in task_list", m, i);
            d.sel_bits |= (uint64_t)src.selected_bits[i] << (4 * i);
        }
        menus[m] = d;
    }
    if (cfg->env_menu)
        for (int i = 0; i < cfg->num_envs; i++)
            if (cfg->env_menu[i] >= cfg->n_menus) return fail(CW_ERR_INVALID, "cw_create: env_menu[%d] out of range", i);

    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(CW_ERR_INVALID, "cw_create: device %d not present (%d devices)", device, ndev);
    DeviceGuard guard(device);
    if (!guard.ok) return fail(CW_ERR_HIP, "cw_create: hipSetDevice(%d) failed", device);

    cw_engine *e = new (std::nothrow) cw_engine();
    if (!e) return fail(CW_ERR_INVALID, "cw_create: out of host memory");
    e->device = device;
    e->obs_mode = cfg->obs_mode;
    e->auto_reset = cfg->auto_reset ? 1 : 0;
    e->n = cfg->num_envs;
    e->S = cfg->size;
    e->ncell = cfg->size * cfg->size;
    e->K = cfg->fixed_init_state;
    e->menus = menus;
    CwParams &P = e->P;
    const size_t N = (size_t)e->n;
    P.n_envs = e->n;
    P.size = e->S;
    P.ncell = e->ncell;
    P.max_steps = cfg->max_steps;
    e->la_period = (unsigned)la_period_for(cfg->max_steps);
    if (const char *v = getenv("CW_TUNE_LA_PERIOD")) if (atoi(v) >= 1) { e->la_period = (unsigned)atoi(v); e->la_adaptive = false; }      // (a forced period: profiles/r06_experiments.txt D)
    e->la_period_max = e->la_period;
    if (const char *v = getenv("CW_TUNE_ROLLOUT_SEGMENT")) e->rollout_segment = atoi(v);
    P.task_mask = (1u << cfg->n_task_list) - 1u;
    P.pool_k = e->K;
    P.div_magic = (uint32_t)((1ull << 32) / (uint64_t)e->S) + 1u;
    P.raster = cfg->raster;
    P.frame_bytes = cfg->raster == CW_RASTER_ALT ? 27u * (uint32_t)e->S * (uint32_t)(e->S + 1) : 48u * (uint32_t)e->ncell;
    {   // Tuning: the defaults are the measured best
        auto geti = [](const char *k, int d) { const char *v = getenv(k); return v ? atoi(v) : d; };
        CwTuning &tn = e->tune;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) tn.n_cu = prop.multiProcessorCount;
        tn.render_chunk_rounds = geti("CW_TUNE_RENDER_CHUNK_ROUNDS", tn.render_chunk_rounds);
        tn.step_envs_per_wave = geti("CW_TUNE_STEP_ENVS_PER_WAVE", tn.step_envs_per_wave);
        if (tn.step_envs_per_wave != 8 && tn.step_envs_per_wave != 16 && tn.step_envs_per_wave != 32) tn.step_envs_per_wave = 64;
        tn.gather = geti("CW_TUNE_GATHER", tn.gather);
        tn.gather_max_size = geti("CW_TUNE_GATHER_MAX_SIZE", tn.gather_max_size);
        if (tn.gather_max_size > 9) tn.gather_max_size = 9;                       // (cw_render_gather_kernel's tables: frames under 4 KiB)
        tn.small_frame_bytes = geti("CW_TUNE_SMALL_FRAME_BYTES", tn.small_frame_bytes);
        tn.small_blocks_per_cu = geti("CW_TUNE_SMALL_BLOCKS", tn.small_blocks_per_cu);
        if (tn.small_blocks_per_cu < 1 || tn.small_blocks_per_cu > 8) tn.small_blocks_per_cu = 1;
        tn.small_launch_bytes = (long long)geti("CW_TUNE_SMALL_LAUNCH_MB", (int)(tn.small_launch_bytes >> 20)) << 20;
        tn.reset_blocks_per_cu = geti("CW_TUNE_RESET_BLOCKS", tn.reset_blocks_per_cu);
        if (tn.reset_blocks_per_cu < 1 || tn.reset_blocks_per_cu > 16) tn.reset_blocks_per_cu = 2;
    }

    int rc = CW_OK;
#define ALLOC(field, count)                                     \
    if (rc == CW_OK) rc = dev_alloc(e, &P.field, (count))
    ALLOC(hdr, N);
    ALLOC(pos, N);
    ALLOC(init_pos, N);
    ALLOC(init_agent, N);
    ALLOC(goal_pos, N);
    ALLOC(goal_codes, N);
    ALLOC(goal_agent, N);
    ALLOC(ep_no, N);
    ALLOC(mt, N * CW_MT_N);
    ALLOC(mt_idx, N);
    ALLOC(pool, N * (size_t)e->K * 9);
    if (rc == CW_OK) rc = dev_alloc(e, &e->seed_scratch, N);
    // step outputs and frames: device memory, or mapped host memory for the single-env loop
#define ALLOC_OUT(field, count)                                 \
    if (rc == CW_OK) rc = cfg->host_outputs ? host_alloc(e, &P.field, (count)) : dev_alloc(e, &P.field, (count))
    ALLOC_OUT(reward, N);
    ALLOC_OUT(done, N);
    ALLOC_OUT(achieved_out, N);
    ALLOC_OUT(desired_out, N);
    ALLOC_OUT(episode_length, N);
    ALLOC_OUT(episode_return, N);
    ALLOC(counters, 8);
    if (rc == CW_OK) {
        // the private stream of the synchronous entry points: at the HIGHEST priority -- the runtime multiplexes streams of one priority onto a handful of
        // hardware queues, and a copy queued behind another engine's thousand pending steps in the same queue waits for them like a device-wide wait
        int pr_least = 0, pr_greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&pr_least, &pr_greatest) != hipSuccess) pr_least = pr_greatest = 0;
        if (hipStreamCreateWithPriority(&e->aux, hipStreamNonBlocking, pr_greatest) != hipSuccess &&
            hipStreamCreateWithFlags(&e->aux, hipStreamNonBlocking) != hipSuccess)
            rc = fail(CW_ERR_HIP, "cw_create: no private stream");
    }
    if (cfg->obs_mode != CW_OBS_STATE) {
        ALLOC_OUT(obs, N * P.frame_bytes);
        ALLOC_OUT(desired_img, N * P.frame_bytes);
        ALLOC_OUT(init_img, N * P.frame_bytes);
        if (cfg->keep_terminal_obs && cfg->auto_reset) ALLOC_OUT(terminal_img, N * P.frame_bytes);
    }
    if (cfg->host_outputs && rc == CW_OK) rc = host_alloc(e, &e->host_actions, N);
    // look-ahead records: every engine that resets by itself and lives in device memory (CW_TUNE_LOOKAHEAD=0: the slow path only, for A/B runs
    // and the test that both give the same results)
    P.lookahead = (cfg->auto_reset && !cfg->host_outputs && !(getenv("CW_TUNE_LOOKAHEAD") && atoi(getenv("CW_TUNE_LOOKAHEAD")) == 0)) ? 1 : 0;
    ALLOC(nx_init_pos, P.lookahead ? N * CW_LA_DEPTH : 1);      // (a queue of CW_LA_DEPTH records per env, one array per slot: cw_layout.h)
    ALLOC(nx_goal_pos, P.lookahead ? N * CW_LA_DEPTH : 1);
    ALLOC(nx_misc, P.lookahead ? N * CW_LA_DEPTH : 1);
    ALLOC(nx_ctl, P.lookahead ? N : 1);
    if (rc == CW_OK && P.lookahead && e->la_adaptive) {      // (no pinned word: a fixed period)
        void *fb = nullptr;
        if (hipHostMalloc(&fb, 64, hipHostMallocCoherent) == hipSuccess) {
            memset(fb, 0, 64);
            e->host_allocs.push_back(fb);
            e->la_feedback = P.la_feedback = (unsigned long long *)fb;
        }
    }
#undef ALLOC_OUT
#undef ALLOC
    CwMenuDev *dmenus = nullptr;
    if (rc == CW_OK) rc = dev_alloc(e, &dmenus, menus.size());
    if (rc == CW_OK && hipMemcpy(dmenus, menus.data(), menus.size() * sizeof(CwMenuDev), hipMemcpyHostToDevice) != hipSuccess)
        rc = fail(CW_ERR_HIP, "cw_create: menu upload failed");
    P.menus = dmenus;
    if (rc == CW_OK) {
        // header: menu id per env; everything else zero until cw_reset
        std::vector<uint32_t> h(N * 4, 0u);
        for (size_t i = 0; i < N; i++) {
            h[i * 4 + 0] = (uint32_t)(cfg->env_menu ? cfg->env_menu[i] : 0) << 24;
            h[i * 4 + 3] = CW_CODES_INITIAL;
        }
        if (hipMemcpy(P.hdr, h.data(), h.size() * 4, hipMemcpyHostToDevice) != hipSuccess)
            rc = fail(CW_ERR_HIP, "cw_create: header upload failed");
    }
    if (rc != CW_OK) {
        if (e->aux) (void)hipStreamDestroy(e->aux);
        for (void *p : e->allocs) (void)hipFree(p);
        for (void *p : e->host_allocs) (void)hipHostFree(p);
        delete e;
        return rc;
    }
    if (cfg->host_outputs && e->n == 1 && !e->auto_reset && e->obs_mode != CW_OBS_PIXELS_FULL) {     // the single-env loop: resident stepper
        void *p = nullptr;
        if (hipHostMalloc(&p, sizeof(CwResident), hipHostMallocCoherent) == hipSuccess &&
            hipStreamCreateWithFlags(&e->res_stream, hipStreamNonBlocking) == hipSuccess) {
            memset(p, 0, sizeof(CwResident));
            e->host_allocs.push_back(p);
            e->res = (CwResident *)p;
            if (host_alloc(e, &e->P.res_onehot, (size_t)e->ncell * 12) != CW_OK) e->P.res_onehot = nullptr;
            int khz = 0;                                 // the kernel's idle-out and time slice are counted on the constant clock
            if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device) == hipSuccess && khz >= 1000) e->res_ticks_per_us = (unsigned long long)khz / 1000ull;
            if (hipEventCreateWithFlags(&e->last_work, hipEventDisableTiming) != hipSuccess) {       // (no event: no resident stepper)
                e->last_work = nullptr;
                e->res = nullptr;
            }
        } else {
            if (p) (void)hipHostFree(p);
            e->res_stream = nullptr;                 // (no resident stepper: cw_step_resident reports it)
        }
    }
    *out = e;
    // default stream: env i seeded like numpy RandomState(i); callers normally reseed
    std::vector<uint32_t> seeds(N);
    for (size_t i = 0; i < N; i++) seeds[i] = (uint32_t)i;
    rc = cw_seed_int(e, seeds.data());
    if (rc == CW_OK) rc = calibrate_sweep(e);
    if (rc != CW_OK) {
        cw_destroy(e);
        *out = nullptr;
    }
    return rc;
}

int cw_destroy(cw_engine *e)
{
    if (!e) return CW_OK;
    DeviceGuard guard(e->device);
    if (!guard.ok) return fail(CW_ERR_HIP, "hipSetDevice(%d) failed", e->device);
    PARK(e);
    (void)resident_park(e);
    (void)quiesce(e);
    prof_free(e);
    if (e->aux) (void)hipStreamDestroy(e->aux);
    if (e->res_stream) (void)hipStreamDestroy(e->res_stream);
    if (e->last_work) (void)hipEventDestroy(e->last_work);
    for (auto &smp : e->guard_ring) for (hipEvent_t ev : smp.ev) if (ev) (void)hipEventDestroy(ev);
    for (void *p : e->allocs) (void)hipFree(p);
    for (void *p : e->host_allocs) (void)hipHostFree(p);
    delete e;
    return CW_OK;
}

// A new stream makes the records computed from the old one void: drop them all; the next refill covers every env.
static hipError_t lookahead_drop(cw_engine *e)
{
    if (!e->P.lookahead) return hipSuccess;
    hipError_t rc = hipMemsetAsync(e->P.nx_misc, 0, (size_t)e->n * CW_LA_DEPTH * sizeof(uint4), e->aux);
    if (rc == hipSuccess) rc = hipMemsetAsync(e->P.nx_ctl, 0, (size_t)e->n * sizeof(uint32_t), e->aux);
    e->la_refill_all = true;
    return rc;
}

// Both seeders run on the device (cw_seed_kernel, one lane per env): the host only copies what the caller handed over.
// (A host-side conversion built an N x 624-word vector serially: 2.6 GB and seconds at 2^20 envs per GPU.)
int cw_seed_mt(cw_engine *e, const uint32_t *keys, const int32_t *pos)
{
    if (!e || !keys || !pos) return fail(CW_ERR_INVALID, "cw_seed_mt: null argument");
    DeviceGuard guard(e->device);
    if (!guard.ok) return fail(CW_ERR_HIP, "hipSetDevice(%d) failed", e->device);
    PARK(e);
    const size_t N = (size_t)e->n;
    for (size_t i = 0; i < N; i++)
        if (pos[i] < 0 || pos[i] > CW_MT_N) return fail(CW_ERR_INVALID, "cw_seed_mt: pos[%zu]=%d outside 0..624", i, pos[i]);
    HIP_TRY(quiesce(e));                              // (the engine's own work, not the card's)
    AuxDrain drain(e);                               // (every exit waits for the private stream: the copies below target this frame's buffers)
    HIP_TRY(aux_copy(e, e->P.mt, keys, N * CW_MT_N * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIP_TRY(aux_copy(e, e->P.mt_idx, pos, N * sizeof(int32_t), hipMemcpyHostToDevice));
    HIP_TRY(cwk_launch_seed(&e->P, nullptr, e->aux));
    HIP_TRY(lookahead_drop(e));
    HIP_TRY(hipStreamSynchronize(e->aux));
    return CW_OK;
}

int cw_seed_int(cw_engine *e, const uint32_t *seeds)
{
    if (!e || !seeds) return fail(CW_ERR_INVALID, "cw_seed_int: null argument");
    DeviceGuard guard(e->device);
    if (!guard.ok) return fail(CW_ERR_HIP, "hipSetDevice(%d) failed", e->device);
    PARK(e);
    HIP_TRY(quiesce(e));
    AuxDrain drain(e);                               // (every exit waits for the private stream: the copies below target this frame's buffers)
    HIP_TRY(aux_copy(e, e->seed_scratch, seeds, (size_t)e->n * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIP_TRY(cwk_launch_seed(&e->P, e->seed_scratch, e->aux));
    HIP_TRY(lookahead_drop(e));
    HIP_TRY(hipStreamSynchronize(e->aux));
    return CW_OK;
}

int cw_get_mt(cw_engine *e, uint32_t *keys, int32_t *pos)
{
    if (!e || !keys || !pos) return fail(CW_ERR_INVALID, "cw_get_mt: null argument");
    DeviceGuard guard(e->device);
    if (!guard.ok) return fail(CW_ERR_HIP, "hipSetDevice(%d) failed", e->device);
    PARK(e);
    const size_t N = (size_t)e->n;
    std::vector<uint32_t> words(N * CW_MT_N);
    std::vector<uint32_t> misc(e->P.lookahead ? N * 4 * CW_LA_DEPTH : 0);
    HIP_TRY(quiesce(e));
    AuxDrain drain(e);                               // (every exit waits for the private stream: the copies below target this frame's buffers)
    HIP_TRY(aux_copy(e, words.data(), e->P.mt, words.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    HIP_TRY(aux_copy(e, pos, e->P.mt_idx, N * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (e->P.lookahead) HIP_TRY(aux_copy(e, misc.data(), e->P.nx_misc, N * 16 * CW_LA_DEPTH, hipMemcpyDeviceToHost));
    HIP_TRY(hipStreamSynchronize(e->aux));
    for (size_t i = 0; i < N; i++) cwh_mt_to_numpy(&words[i * CW_MT_N], pos[i], keys + i * CW_MT_N);
    if (e->P.lookahead) {                            // the engine's streams are as many resets ahead as records wait: report the position BEFORE the first of them
        for (size_t i = 0; i < N; i++) {
            uint32_t ahead = 0;
            for (size_t d = 0; d < CW_LA_DEPTH; d++)
                if (misc[(d * N + i) * 4 + 2] >> 31) ahead += misc[(d * N + i) * 4 + 3];
            if (ahead) cwh_mt_rewind(keys + i * CW_MT_N, &pos[i], ahead);
        }
    }
    // numpy's own form of a stream that stands at a generation's end: it regenerates lazily, so after the 624th draw RandomState.get_state() shows
    // (the generation just used up, 624), never (the next one, 0) -- the engine's consume-and-replace form holds the next one already
    for (size_t i = 0; i < N; i++)
        if (pos[i] == 0) { cwh_mt_untwist(keys + i * CW_MT_N); pos[i] = CW_MT_N; }
    return CW_OK;
}

int cw_generate_fixed_states(cw_engine *e, cw_stream_t stream)
{
    if (!e) return fail(CW_ERR_INVALID, "cw_generate_fixed_states: null engine");
    if (e->K == 0) return CW_OK;
    DeviceGuard guard(e->device);
    if (!guard.ok) return fail(CW_ERR_HIP, "hipSetDevice(%d) failed", e->device);
    PARK(e);
    if (e->P.lookahead && e->has_reset) {            // records wait, computed from the stream the pool is about to draw from: rewind and drop them
        std::vector<uint32_t> keys((size_t)e->n * CW_MT_N);
        std::vector<int32_t> pos((size_t)e->n);
        int rc = cw_get_mt(e, keys.data(), pos.data());
        if (rc == CW_OK) rc = cw_seed_mt(e, keys.data(), pos.data());
        if (rc != CW_OK) return rc;
    }
    HIP_TRY(cwk_launch_pool(&e->P, &e->tune, (hipStream_t)stream));
    // one-time, off the hot path: the pool and the advanced RNG streams are complete before any other stream can reset from them
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return CW_OK;
}

int cw_reset(cw_engine *e, cw_stream_t stream)
{
    if (!e) return fail(CW_ERR_INVALID, "cw_reset: null engine");
    DeviceGuard guard(e->device);
    if (!guard.ok) return fail(CW_ERR_HIP, "hipSetDevice(%d) failed", e->device);
    PARK(e);
    note_work(e, (hipStream_t)stream);
    HIP_TRY(cwk_launch_reset_all(&e->P, &e->tune, e->obs_mode, (hipStream_t)stream));
    if (e->P.lookahead) {                            // every env's NEXT episode, ahead of time (cw_refill_kernel)
        HIP_TRY(cwk_launch_refill(&e->P, &e->tune, 1, (hipStream_t)stream));
        e->la_refill_all = false;
        e->la_steps = 0;
    }
    e->has_reset = true;
    if (e->res && hipEventRecord(e->last_work, (hipStream_t)stream) == hipSuccess) e->last_work_set = true;
    return CW_OK;
}

int cw_step(cw_engine *e, const void *actions, int action_dtype, cw_stream_t stream)
{
    if (!e || !actions) return fail(CW_ERR_INVALID, "cw_step: null argument");
    if (action_dtype < CW_ACT_I32 || action_dtype > CW_ACT_U8) return fail(CW_ERR_INVALID, "cw_step: bad action dtype %d", action_dtype);
    if (!e->has_reset) return fail(CW_ERR_STATE, "cw_step called before cw_reset");
    DeviceGuard guard(e->device);
    if (!guard.ok) return fail(CW_ERR_HIP, "hipSetDevice(%d) failed", e->device);
    PARK(e);
    const bool capturing = e->in_step_many ? e->capturing_now : stream_capturing((hipStream_t)stream);
    if (capturing) e->captured = true;
    else if (e->n_work != 1 || e->work[0] != (hipStream_t)stream) note_work(e, (hipStream_t)stream);
    hipEvent_t *ev = (e->prof_n < e->prof_cap) ? &e->prof_ev[(size_t)e->prof_n * 6] : nullptr;
    const bool profiled = ev != nullptr;
    if (e->guard_on && !ev && !capturing) ev = sweep_guard_tick(e, (hipStream_t)stream);       // (a captured graph keeps the rate it was captured with)
    // a step captured into a HIP graph on its own carries the refill with it: a replayed graph would otherwise never refill
    if (capturing && e->P.lookahead && !e->la_refill_all && e->la_steps + 1 < e->la_period && !e->in_step_many) e->la_steps = e->la_period;
    if (e->P.lookahead && (e->la_refill_all || ++e->la_steps >= e->la_period)) {      // look-ahead refill, between two steps
        if (e->la_feedback && !capturing) la_adapt(e);
        HIP_TRY(cwk_launch_refill(&e->P, &e->tune, e->la_refill_all ? 1 : 0, (hipStream_t)stream));
        e->la_refill_all = false;
        e->la_steps = 0;
    }
    HIP_TRY(cwk_launch_step(&e->P, &e->tune, actions, action_dtype, e->obs_mode, e->auto_reset, (hipStream_t)stream, ev));
    if (profiled) e->prof_n++;
    if (e->res && hipEventRecord(e->last_work, (hipStream_t)stream) == hipSuccess) e->last_work_set = true;
    return CW_OK;
}

// n_steps consecutive steps from an action array [n_steps][N]: what a loop over cw_step enqueues, without the caller's per-step cost (a Python
// loop spends more per step than a state-only step takes on the card).  Capturable into a HIP graph as one piece; inside a capture the
// look-ahead refill rides at the head of the sequence too, so that a replayed graph of fewer than e->la_period steps still refills.
int cw_step_many(cw_engine *e, const void *actions, int action_dtype, int32_t n_steps, cw_stream_t stream)
{
    if (!e || !actions) return fail(CW_ERR_INVALID, "cw_step_many: null argument");
    if (n_steps < 1) return fail(CW_ERR_INVALID, "cw_step_many: n_steps must be >= 1");
    if (action_dtype < CW_ACT_I32 || action_dtype > CW_ACT_U8) return fail(CW_ERR_INVALID, "cw_step_many: bad action dtype %d", action_dtype);
    e->capturing_now = stream_capturing((hipStream_t)stream);
    if (e->P.lookahead && e->capturing_now) e->la_steps = e->la_period;
    const size_t row = (size_t)e->n * (action_dtype == CW_ACT_I32 ? 4 : action_dtype == CW_ACT_I64 ? 8 : 1);
    e->in_step_many = true;
    int rc = CW_OK;
    for (int32_t t = 0; t < n_steps && rc == CW_OK; t++) rc = cw_step(e, (const unsigned char *)actions + (size_t)t * row, action_dtype, stream);
    e->in_step_many = false;
    return rc;
}

// One step of the single-env loop WITHOUT a kernel launch: ring the resident kernel's doorbell, spin on its answer (cw_kernels.hip:
// cw_resident_kernel).  The kernel is (re)launched on demand -- the first call, after it idled out (0.5 ms without a request), after its
// time slice (200 ms), after any other entry point parked it -- and a request that raced with its exit is served by the next instance:
// `ack` says which sequence number was served last, and a new instance starts from there.
int cw_step_resident(cw_engine *e, int32_t action, int32_t want_onehot)
{
    if (!e) return fail(CW_ERR_INVALID, "cw_step_resident: null engine");
    if (!e->res) return fail(CW_ERR_INVALID, "cw_step_resident needs num_envs == 1, host_outputs, auto_reset == 0 and obs_mode state or pixels_dirty");
    if (!e->has_reset) return fail(CW_ERR_STATE, "cw_step_resident called before cw_reset");
    if (action < 0 || action > 127) return fail(CW_ERR_INVALID, "cw_step_resident: action %d outside 0..127", action);
    if (want_onehot && !e->P.res_onehot) return fail(CW_ERR_INVALID, "cw_step_resident: no host one-hot buffer on this engine");
    CwResident *R = e->res;
    const uint32_t seq = (++e->res_seq) & 0xFFFFFFu;
    __atomic_store_n(&R->bell, (unsigned long long)((seq << 8) | (want_onehot ? 0x80u : 0u) | (uint32_t)action), __ATOMIC_RELEASE);
    struct timespec t0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (unsigned spins = 0;; spins++) {
        if (e->res_running && __atomic_load_n(&R->ack, __ATOMIC_ACQUIRE) == seq) return CW_OK;
        if (!e->res_running || __atomic_load_n(&R->exited, __ATOMIC_ACQUIRE) != 0) {
            // no kernel on the card, or it has left (idle, time slice): wait for it to be gone, then start one that continues after `ack`
            DeviceGuard guard(e->device);
            if (!guard.ok) return fail(CW_ERR_HIP, "hipSetDevice(%d) failed", e->device);
            if (e->res_running) HIP_TRY(hipStreamSynchronize(e->res_stream));
            e->res_running = false;
            if (__atomic_load_n(&R->ack, __ATOMIC_ACQUIRE) == seq) {            // it answered on its way out
                __atomic_store_n(&R->exited, 0u, __ATOMIC_RELEASE);
                return CW_OK;
            }
            // (the new instance reads the env's records: whatever cw_reset / cw_step enqueued last comes first -- an event of the engine's own,
            // recorded when that work was enqueued: the caller's stream may be gone by now)
            if (e->last_work_set) { HIP_TRY(hipStreamWaitEvent(e->res_stream, e->last_work, 0)); e->last_work_set = false; }
            __atomic_store_n(&R->exited, 0u, __ATOMIC_RELEASE);
            __atomic_store_n(&R->ack, (seq - 1u) & 0xFFFFFFu, __ATOMIC_RELEASE);
            HIP_TRY(cwk_launch_resident(&e->P, R, (seq - 1u) & 0xFFFFFFu, e->obs_mode == CW_OBS_PIXELS_DIRTY ? 1 : 0,
                                        500ull * e->res_ticks_per_us /* 0.5 ms idle */, 200000ull * e->res_ticks_per_us /* 200 ms slice */, e->res_stream));
            e->res_running = true;
            continue;
        }
        __builtin_ia32_pause();
        if ((spins & 1023u) == 1023u) {
            struct timespec t1;
            clock_gettime(CLOCK_MONOTONIC, &t1);
            if ((t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec) > 5.0) {
                // WITHDRAW the request before parking: the kernel serves a pending sequence number before it honours the stop bit, so the doorbell
                // goes back to the last number it answered, with the stop bit -- and if it did serve the step meanwhile, that is the answer
                const uint32_t acked = __atomic_load_n(&R->ack, __ATOMIC_ACQUIRE);
                __atomic_store_n(&R->bell, (1ull << 32) | ((unsigned long long)acked << 8), __ATOMIC_RELEASE);
                const hipError_t he = hipStreamSynchronize(e->res_stream);
                e->res_running = false;
                const uint32_t now_acked = __atomic_load_n(&R->ack, __ATOMIC_ACQUIRE);
                __atomic_store_n(&R->bell, (unsigned long long)now_acked << 8, __ATOMIC_RELEASE);
                __atomic_store_n(&R->exited, 0u, __ATOMIC_RELEASE);
                e->res_seq = now_acked;
                if (he == hipSuccess && now_acked == seq) return CW_OK;
                return fail(CW_ERR_HIP, "cw_step_resident: no answer from the resident kernel within 5 s (request withdrawn, kernel parked: the step was not taken)");
            }
        }
    }
}

int cw_resident_stop(cw_engine *e)
{
    if (!e) return fail(CW_ERR_INVALID, "cw_resident_stop: null engine");
    DeviceGuard guard(e->device);
    if (!guard.ok) return fail(CW_ERR_HIP, "hipSetDevice(%d) failed", e->device);
    return resident_park(e);
}

int cw_rollout(cw_engine *e, const uint8_t *actions, int32_t n_steps, int32_t *rewards, uint8_t *dones, cw_stream_t stream)
{
    if (!e || !actions) return fail(CW_ERR_INVALID, "cw_rollout: null argument");
    if (n_steps < 1) return fail(CW_ERR_INVALID, "cw_rollout: n_steps must be >= 1");
    if (e->obs_mode != CW_OBS_STATE || !e->auto_reset)
        return fail(CW_ERR_INVALID, "cw_rollout needs obs_mode CW_OBS_STATE and auto_reset (frames are not painted by the persistent kernel)");
    if (!e->has_reset) return fail(CW_ERR_STATE, "cw_rollout called before cw_reset");
    DeviceGuard guard(e->device);
    if (!guard.ok) return fail(CW_ERR_HIP, "hipSetDevice(%d) failed", e->device);
    PARK(e);
    if (stream_capturing((hipStream_t)stream)) e->captured = true; else note_work(e, (hipStream_t)stream);
    // Engines with look-ahead records: the n_steps go out in SEGMENTS of max_steps steps (64 at least), a refill kernel ahead of each -- a persistent launch
    // cannot refill, and an env that finishes a second time inside it finds no record and is reset the slow way by its whole wave (~12 us per env: the second
    // all-env time-out of a 600-step launch cost 64 envs x 12 us per wave, as much as the 600 steps themselves).  An episode lasts max_steps steps at most, so
    // with one segment per max_steps steps every time-out finds its record; shorter segments cost more in launches than they save (a segment start is ~20 us:
    // the refill, the launch, the state's round trip): 65 536 envs, T = 600: one launch 2.2e10 env-steps/s, segments of 64 / 128 / 300 steps 2.0 / 2.7 / 2.9e10
    // (2^20 envs: 4.1 -> 6.0e10; profiles/r06_experiments.txt C).  Same results, the same stream order.  CW_TUNE_ROLLOUT_SEGMENT=n: segments of n steps, 0: one launch.
    const int32_t seg = !e->P.lookahead || e->rollout_segment == 0 ? n_steps : e->rollout_segment > 0 ? e->rollout_segment : (e->P.max_steps > 64 ? e->P.max_steps : 64);
    const size_t N = (size_t)e->n;
    for (int32_t t0 = 0; t0 < n_steps; t0 += seg) {
        if (e->P.lookahead) {
            HIP_TRY(cwk_launch_refill(&e->P, &e->tune, e->la_refill_all ? 1 : 0, (hipStream_t)stream));
            e->la_refill_all = false;
            e->la_steps = 0;
        }
        const int32_t n = n_steps - t0 < seg ? n_steps - t0 : seg;
        HIP_TRY(cwk_launch_rollout(&e->P, actions + (size_t)t0 * N, n, rewards ? rewards + (size_t)t0 * N : nullptr, dones ? dones + (size_t)t0 * N : nullptr,
                                   (hipStream_t)stream));
    }
    return CW_OK;
}

int cw_render(cw_engine *e, uint8_t *out_frames, cw_stream_t stream)
{
    if (!e || !out_frames) return fail(CW_ERR_INVALID, "cw_render: null argument");
    if (!e->has_reset) return fail(CW_ERR_STATE, "cw_render called before cw_reset");
    DeviceGuard guard(e->device);
    if (!guard.ok) return fail(CW_ERR_HIP, "hipSetDevice(%d) failed", e->device);
    PARK(e);
    note_work(e, (hipStream_t)stream);
    HIP_TRY(cwk_launch_render_ext(&e->P, &e->tune, out_frames, (hipStream_t)stream));
    return CW_OK;
}

int cw_render_onehot(cw_engine *e, const uint8_t *onehot, int32_t n_states, uint16_t *out_frames, cw_stream_t stream)
{
    if (!e || !onehot || !out_frames) return fail(CW_ERR_INVALID, "cw_render_onehot: null argument");
    if (n_states < 1) return fail(CW_ERR_INVALID, "cw_render_onehot: n_states must be >= 1");
    DeviceGuard guard(e->device);
    if (!guard.ok) return fail(CW_ERR_HIP, "hipSetDevice(%d) failed", e->device);
    PARK(e);
    note_work(e, (hipStream_t)stream);
    HIP_TRY(cwk_launch_render_onehot(&e->P, onehot, n_states, out_frames, (hipStream_t)stream));
    return CW_OK;
}

int cw_export_grid(cw_engine *e, uint8_t *out, cw_stream_t stream)
{
    if (!e || !out) return fail(CW_ERR_INVALID, "cw_export_grid: null argument");
    DeviceGuard guard(e->device);
    if (!guard.ok) return fail(CW_ERR_HIP, "hipSetDevice(%d) failed", e->device);
    PARK(e);
    note_work(e, (hipStream_t)stream);
    HIP_TRY(cwk_launch_export(&e->P, &e->tune, out, 0, 0, (hipStream_t)stream));
    return CW_OK;
}

int cw_export_onehot(cw_engine *e, uint8_t *out, cw_stream_t stream)
{
    if (!e || !out) return fail(CW_ERR_INVALID, "cw_export_onehot: null argument");
    DeviceGuard guard(e->device);
    if (!guard.ok) return fail(CW_ERR_HIP, "hipSetDevice(%d) failed", e->device);
    PARK(e);
    note_work(e, (hipStream_t)stream);
    HIP_TRY(cwk_launch_export(&e->P, &e->tune, out, 1, 0, (hipStream_t)stream));
    return CW_OK;
}

int cw_export_onehot_of(cw_engine *e, int which, uint8_t *out, cw_stream_t stream)
{
    if (!e || !out) return fail(CW_ERR_INVALID, "cw_export_onehot_of: null argument");
    if (which < CW_STATE_CURRENT || which > CW_STATE_INIT) return fail(CW_ERR_INVALID, "cw_export_onehot_of: which must be CW_STATE_*");
    if (!e->has_reset) return fail(CW_ERR_STATE, "cw_export_onehot_of called before cw_reset");
    DeviceGuard guard(e->device);
    if (!guard.ok) return fail(CW_ERR_HIP, "hipSetDevice(%d) failed", e->device);
    PARK(e);
    note_work(e, (hipStream_t)stream);
    HIP_TRY(cwk_launch_export(&e->P, &e->tune, out, 1, which, (hipStream_t)stream));
    return CW_OK;
}

int cw_profile_begin(cw_engine *e, int max_steps)
{
    if (!e || max_steps < 1 || max_steps > 100000) return fail(CW_ERR_INVALID, "cw_profile_begin: bad argument");
    DeviceGuard guard(e->device);
    if (!guard.ok) return fail(CW_ERR_HIP, "hipSetDevice(%d) failed", e->device);
    PARK(e);
    prof_free(e);
    e->prof_ev.resize((size_t)max_steps * 6);
    for (auto &ev : e->prof_ev) HIP_TRY(hipEventCreate(&ev));
    e->prof_cap = max_steps;
    return CW_OK;
}

int cw_profile_end(cw_engine *e, cw_profile *out)
{
    if (!e || !out) return fail(CW_ERR_INVALID, "cw_profile_end: null argument");
    DeviceGuard guard(e->device);
    if (!guard.ok) return fail(CW_ERR_HIP, "hipSetDevice(%d) failed", e->device);
    PARK(e);
    memset(out, 0, sizeof(*out));
    const int n = e->prof_n;
    if (n > 0) {
        const bool side_recorded = e->obs_mode != CW_OBS_PIXELS_FULL;      // (full-frame mode: only the sweep is bracketed)
        for (int k = 0; k < 6; k++)
            if (side_recorded || k >= 4) HIP_TRY(hipEventSynchronize(e->prof_ev[(size_t)(n - 1) * 6 + k]));
        double acc[3] = {0, 0, 0};
        float rmax = 0.f, rmin = 1e30f;
        std::vector<float> render_ms;
        render_ms.reserve((size_t)n);
        for (int i = 0; i < n; i++)
            for (int k = 0; k < 3; k++) {
                if (k != 2 && !side_recorded) continue;      // overlapped full-pixel step: only the render kernel is bracketed
                float ms = 0.f;
                HIP_TRY(hipEventElapsedTime(&ms, e->prof_ev[(size_t)i * 6 + 2 * k], e->prof_ev[(size_t)i * 6 + 2 * k + 1]));
                acc[k] += ms;
                if (k == 2) { rmax = ms > rmax ? ms : rmax; rmin = ms < rmin ? ms : rmin; render_ms.push_back(ms); }
            }
        out->steps = n;
        out->ms_step_kernel = (float)(acc[0] / n);
        out->ms_reset_kernel = (float)(acc[1] / n);
        out->ms_render_kernel = (float)(acc[2] / n);
        out->ms_render_kernel_max = rmax;
        out->ms_render_kernel_min = rmin;
        std::sort(render_ms.begin(), render_ms.end());
        out->ms_render_kernel_median = render_ms.empty() ? 0.f : render_ms[render_ms.size() / 2];
    }
    prof_free(e);
    return CW_OK;
}

const char *cw_render_kernel_name(const cw_engine *e)
{
    if (!e || e->obs_mode == CW_OBS_STATE) return "";
    if (e->obs_mode == CW_OBS_PIXELS_DIRTY) return e->auto_reset ? "cw_step_fused_kernel" : "cw_step_kernel";
    if (e->tune.gather && e->P.raster == CW_RASTER_RAY && e->S <= e->tune.gather_max_size) return "cw_render_gather_kernel";
    return "cw_render_pieces_kernel";
}

int cw_tuner(const cw_engine *e, cw_tuner_state *out)
{
    if (!e || !out) return fail(CW_ERR_INVALID, "cw_tuner: null argument");
    out->period16 = e->tune.period16;
    out->period16_head = e->tune.period16_head;
    out->period16_busy = e->tune.period16_busy;
    out->lookahead = e->P.lookahead;
    out->resident = e->res ? 1 : 0;
    out->guard_slowdowns = e->guard_on ? e->guard.slowdowns : -1;
    return CW_OK;
}

int cw_buffers(cw_engine *e, cw_buffer_table *out)
{
    if (!e || !out) return fail(CW_ERR_INVALID, "cw_buffers: null argument");
    const CwParams &P = e->P;
    out->obs = P.obs;
    out->desired_goal = P.desired_img;
    out->init_obs = P.init_img;
    out->terminal_obs = P.terminal_img;
    out->reward = P.reward;
    out->done = P.done;
    out->achieved = P.achieved_out;
    out->desired = P.desired_out;
    out->episode_length = P.episode_length;
    out->episode_return = P.episode_return;
    out->hdr = (uint8_t *)P.hdr;
    out->slot_pos = (uint16_t *)P.pos;
    out->counters = (uint64_t *)P.counters;
    out->frame_bytes = P.frame_bytes;
    out->host_actions = e->host_actions;
    out->host_onehot = P.res_onehot;
    return CW_OK;
}

int cw_synchronize(cw_engine *e, cw_stream_t stream)
{
    if (!e) return fail(CW_ERR_INVALID, "cw_synchronize: null engine");
    DeviceGuard guard(e->device);
    if (!guard.ok) return fail(CW_ERR_HIP, "hipSetDevice(%d) failed", e->device);
    PARK(e);
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    for (int i = 0; i < e->n_work; i++)               // (that stream's share of the engine's work is done: the caller may destroy it now)
        if (e->work[i] == (hipStream_t)stream) { e->work[i] = e->work[--e->n_work]; break; }
    return CW_OK;
}

// generate_fixed_states' result (ray.py:116-118: fixed_state_list): the K pooled placements of every env as cell indices, [N][K][9] uint16 =
// objects 0..7 (OBJECTS order, ray.py:21) then the agent.  Synchronous host call.
int cw_get_fixed_states(cw_engine *e, uint16_t *out)
{
    if (!e || !out) return fail(CW_ERR_INVALID, "cw_get_fixed_states: null argument");
    if (e->K == 0) return fail(CW_ERR_INVALID, "cw_get_fixed_states: the engine was created with fixed_init_state = 0");
    DeviceGuard guard(e->device);
    if (!guard.ok) return fail(CW_ERR_HIP, "hipSetDevice(%d) failed", e->device);
    PARK(e);
    HIP_TRY(quiesce(e));
    AuxDrain drain(e);                               // (every exit waits for the private stream: the copies below target this frame's buffers)
    HIP_TRY(aux_copy(e, out, e->P.pool, (size_t)e->n * e->K * 9 * sizeof(uint16_t), hipMemcpyDeviceToHost));
    HIP_TRY(hipStreamSynchronize(e->aux));
    return CW_OK;
}

// ------------------------------------------------------------------------------ state get/set
int cw_get_state(cw_engine *e, cw_state_view *v)
{
    if (!e || !v) return fail(CW_ERR_INVALID, "cw_get_state: null argument");
    DeviceGuard guard(e->device);
    if (!guard.ok) return fail(CW_ERR_HIP, "hipSetDevice(%d) failed", e->device);
    PARK(e);
    const size_t N = (size_t)e->n;
    const int S = e->S, nc = e->ncell;
    std::vector<uint32_t> hdr(N * 4), goal_codes(N);
    std::vector<uint16_t> pos(N * 8), ipos(N * 8), gpos(N * 8), iagent(N), gagent(N);
    std::vector<int32_t> epno(N);
    HIP_TRY(quiesce(e));                              // (the engine's own work; copies on its private stream)
    AuxDrain drain(e);                               // (every exit waits for the private stream: the copies below target this frame's buffers)
    HIP_TRY(aux_copy(e, hdr.data(), e->P.hdr, N * 16, hipMemcpyDeviceToHost));
    HIP_TRY(aux_copy(e, pos.data(), e->P.pos, N * 16, hipMemcpyDeviceToHost));
    HIP_TRY(aux_copy(e, ipos.data(), e->P.init_pos, N * 16, hipMemcpyDeviceToHost));
    HIP_TRY(aux_copy(e, gpos.data(), e->P.goal_pos, N * 16, hipMemcpyDeviceToHost));
    HIP_TRY(aux_copy(e, goal_codes.data(), e->P.goal_codes, N * 4, hipMemcpyDeviceToHost));
    HIP_TRY(aux_copy(e, iagent.data(), e->P.init_agent, N * 2, hipMemcpyDeviceToHost));
    HIP_TRY(aux_copy(e, gagent.data(), e->P.goal_agent, N * 2, hipMemcpyDeviceToHost));
    HIP_TRY(aux_copy(e, epno.data(), e->P.ep_no, N * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipStreamSynchronize(e->aux));
    for (size_t i = 0; i < N; i++) {
        const uint32_t *h = &hdr[i * 4];
        if (v->grid) cwh_slots_to_grid(&pos[i * 8], h[3], nc, v->grid + i * nc);
        if (v->init_grid) cwh_slots_to_grid(&ipos[i * 8], CW_CODES_INITIAL, nc, v->init_grid + i * nc);
        if (v->goal_grid) cwh_slots_to_grid(&gpos[i * 8], goal_codes[i], nc, v->goal_grid + i * nc);
        if (v->agent_rc) { v->agent_rc[i * 2] = h[0] & 0xFF; v->agent_rc[i * 2 + 1] = (h[0] >> 8) & 0xFF; }
        if (v->init_agent_rc) { v->init_agent_rc[i * 2] = (uint8_t)(iagent[i] / S); v->init_agent_rc[i * 2 + 1] = (uint8_t)(iagent[i] % S); }
        if (v->goal_agent_rc) { v->goal_agent_rc[i * 2] = (uint8_t)(gagent[i] / S); v->goal_agent_rc[i * 2 + 1] = (uint8_t)(gagent[i] % S); }
        if (v->hold) v->hold[i] = (h[0] >> 16) & 0xFF;
        if (v->achieved) v->achieved[i] = (uint16_t)(h[1] & 0xFFFF);
        if (v->desired) v->desired[i] = (uint16_t)(h[1] >> 16);
        if (v->step_num) v->step_num[i] = (int32_t)(h[2] & 0xFFFF);
        if (v->ep_no) v->ep_no[i] = epno[i];
    }
    return CW_OK;
}

int cw_set_state(cw_engine *e, const cw_state_view *v)
{
    if (!e || !v) return fail(CW_ERR_INVALID, "cw_set_state: null argument");
    if (!e->has_reset) return fail(CW_ERR_STATE, "cw_set_state called before cw_reset");
    DeviceGuard guard(e->device);
    if (!guard.ok) return fail(CW_ERR_HIP, "hipSetDevice(%d) failed", e->device);
    PARK(e);
    const size_t N = (size_t)e->n;
    const int S = e->S, nc = e->ncell;
    std::vector<uint32_t> hdr(N * 4);
    std::vector<uint16_t> pos(N * 8), ipos(N * 8), gpos, gagent, iagent;
    std::vector<uint32_t> gcodes;
    std::vector<int32_t> epno(N);
    const bool restore_episode = v->goal_grid || v->goal_agent_rc || v->init_agent_rc;
    HIP_TRY(quiesce(e));
    AuxDrain drain(e);                               // (every exit waits for the private stream: the copies below target this frame's buffers)
    if (restore_episode) {                           // the episode records: goal state (imagine_obs' result) and the agent's start cell
        gpos.resize(N * 8); gagent.resize(N); iagent.resize(N); gcodes.resize(N);
        HIP_TRY(aux_copy(e, gpos.data(), e->P.goal_pos, N * 16, hipMemcpyDeviceToHost));
        HIP_TRY(aux_copy(e, gcodes.data(), e->P.goal_codes, N * 4, hipMemcpyDeviceToHost));
        HIP_TRY(aux_copy(e, gagent.data(), e->P.goal_agent, N * 2, hipMemcpyDeviceToHost));
        HIP_TRY(aux_copy(e, iagent.data(), e->P.init_agent, N * 2, hipMemcpyDeviceToHost));
    }
    HIP_TRY(aux_copy(e, hdr.data(), e->P.hdr, N * 16, hipMemcpyDeviceToHost));
    HIP_TRY(aux_copy(e, pos.data(), e->P.pos, N * 16, hipMemcpyDeviceToHost));
    HIP_TRY(aux_copy(e, ipos.data(), e->P.init_pos, N * 16, hipMemcpyDeviceToHost));
    HIP_TRY(aux_copy(e, epno.data(), e->P.ep_no, N * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipStreamSynchronize(e->aux));
    for (size_t i = 0; i < N; i++) {
        uint32_t *h = &hdr[i * 4];
        uint32_t hold = (h[0] >> 16) & 0xFF;
        if (v->hold) {
            if (v->hold[i] > 3) return fail(CW_ERR_INVALID, "cw_set_state: hold[%zu]=%d outside 0..3", i, v->hold[i]);
            hold = v->hold[i];
        }
        if (v->agent_rc) {
            if (v->agent_rc[i * 2] >= S || v->agent_rc[i * 2 + 1] >= S) return fail(CW_ERR_INVALID, "cw_set_state: agent of env %zu off the grid", i);
            h[0] = (h[0] & 0xFFFF0000u) | v->agent_rc[i * 2] | ((uint32_t)v->agent_rc[i * 2 + 1] << 8);
        }
        if (v->grid) {
            // every object on the grid takes a slot; the held object (if any) takes one more
            uint32_t codes = 0;
            int k = 0;
            const uint8_t *g = v->grid + i * nc;
            for (int c = 0; c < nc; c++) {
                if (g[c] == 0) continue;
                if (g[c] > 8) return fail(CW_ERR_INVALID, "cw_set_state: env %zu cell %d has code %d > 8", i, c, g[c]);
                if (k >= 8) return fail(CW_ERR_INVALID, "cw_set_state: env %zu has more than 8 objects (outside the reference's reachable states)", i);
                pos[i * 8 + k] = (uint16_t)c;
                codes |= (uint32_t)g[c] << (4 * k);
                k++;
            }
            if (hold) {
                if (k >= 8) return fail(CW_ERR_INVALID, "cw_set_state: env %zu: 8 objects on the grid plus a held one", i);
                pos[i * 8 + k] = CW_POS_HELD;
                codes |= hold << (4 * k);
                k++;
            }
            for (; k < 8; k++) pos[i * 8 + k] = CW_POS_GONE;
            h[3] = codes;
        } else if (v->hold) {
            return fail(CW_ERR_INVALID, "cw_set_state: hold given without grid");
        }
        h[0] = (h[0] & 0xFF00FFFFu) | (hold << 16);
        if (v->init_grid) {
            const uint8_t *g = v->init_grid + i * nc;
            for (int k = 0; k < 8; k++) ipos[i * 8 + k] = CW_POS_GONE;
            for (int c = 0; c < nc; c++) {
                if (g[c] == 0) continue;
                if (g[c] > 8) return fail(CW_ERR_INVALID, "cw_set_state: env %zu init cell %d has code %d > 8", i, c, g[c]);
                if (ipos[i * 8 + g[c] - 1] != CW_POS_GONE)
                    return fail(CW_ERR_INVALID, "cw_set_state: env %zu init grid holds object %d twice (sample_state places one of each, ray.py:605-608)", i, g[c]);
                ipos[i * 8 + g[c] - 1] = (uint16_t)c;
            }
        }
        if (v->goal_grid) {                          // any slot order paints the same goal frame
            const uint8_t *g = v->goal_grid + i * nc;
            uint32_t codes = 0;
            int k = 0;
            for (int c = 0; c < nc; c++) {
                if (g[c] == 0) continue;
                if (g[c] > 8 || k >= 8) return fail(CW_ERR_INVALID, "cw_set_state: env %zu goal grid: code > 8 or more than 8 objects", i);
                gpos[i * 8 + k] = (uint16_t)c;
                codes |= (uint32_t)g[c] << (4 * k);
                k++;
            }
            for (; k < 8; k++) gpos[i * 8 + k] = CW_POS_GONE;
            gcodes[i] = codes;
        }
        if (v->goal_agent_rc) {
            if (v->goal_agent_rc[i * 2] >= S || v->goal_agent_rc[i * 2 + 1] >= S) return fail(CW_ERR_INVALID, "cw_set_state: goal agent of env %zu off the grid", i);
            gagent[i] = (uint16_t)(v->goal_agent_rc[i * 2] * S + v->goal_agent_rc[i * 2 + 1]);
        }
        if (v->init_agent_rc) {
            if (v->init_agent_rc[i * 2] >= S || v->init_agent_rc[i * 2 + 1] >= S) return fail(CW_ERR_INVALID, "cw_set_state: init agent of env %zu off the grid", i);
            iagent[i] = (uint16_t)(v->init_agent_rc[i * 2] * S + v->init_agent_rc[i * 2 + 1]);
        }
        if (v->achieved) h[1] = (h[1] & 0xFFFF0000u) | v->achieved[i];
        if (v->desired) h[1] = (h[1] & 0x0000FFFFu) | ((uint32_t)v->desired[i] << 16);
        if (v->step_num) {
            if (v->step_num[i] < 0 || v->step_num[i] > 65535) return fail(CW_ERR_INVALID, "cw_set_state: step_num[%zu] outside 0..65535", i);
            h[2] = (h[2] & 0xFFFF0000u) | (uint32_t)v->step_num[i];
        }
        if (v->ep_no) epno[i] = v->ep_no[i];
    }
    HIP_TRY(aux_copy(e, e->P.hdr, hdr.data(), N * 16, hipMemcpyHostToDevice));
    HIP_TRY(aux_copy(e, e->P.pos, pos.data(), N * 16, hipMemcpyHostToDevice));
    HIP_TRY(aux_copy(e, e->P.init_pos, ipos.data(), N * 16, hipMemcpyHostToDevice));
    HIP_TRY(aux_copy(e, e->P.ep_no, epno.data(), N * 4, hipMemcpyHostToDevice));
    if (restore_episode) {
        HIP_TRY(aux_copy(e, e->P.goal_pos, gpos.data(), N * 16, hipMemcpyHostToDevice));
        HIP_TRY(aux_copy(e, e->P.goal_codes, gcodes.data(), N * 4, hipMemcpyHostToDevice));
        HIP_TRY(aux_copy(e, e->P.goal_agent, gagent.data(), N * 2, hipMemcpyHostToDevice));
        HIP_TRY(aux_copy(e, e->P.init_agent, iagent.data(), N * 2, hipMemcpyHostToDevice));
    }
    if (e->obs_mode != CW_OBS_STATE) {   // the persistent frames must follow the injected state
        if (restore_episode) HIP_TRY(cwk_launch_render_restore(&e->P, &e->tune, e->aux));
        else HIP_TRY(cwk_launch_render_ext(&e->P, &e->tune, e->P.obs, e->aux));
    }
    HIP_TRY(hipStreamSynchronize(e->aux));
    return CW_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------ checkpoint blob
// The engine's complete resumable state as one opaque host blob: the RAW device records (header, slots, episode
// records, RNG streams in engine form, fixed_init_state pool, last step outputs, counters), restored verbatim -- so a
// resumed engine returns the same tensors as the run that never stopped, slot order included -- plus a header that
// pins the configuration the records only make sense under (batch, grid, episode length, task_list width, pool size,
// and a hash of the task menus; each env's menu id and reward rule travel inside its header record).
struct CwCkptHeader {
    char magic[8];
    uint32_t version, header_bytes;
    int32_t n_envs, size, max_steps, pool_k;
    uint32_t task_mask, n_menus, lookahead, reserved;
    uint64_t menus_hash, total_bytes;
};
static const char CW_CKPT_MAGIC[8] = {'C', 'W', 'C', 'K', 'P', 'T', 0, 1};
enum { CW_CKPT_VERSION = 4 };      // 2: look-ahead records; 3: episode_return, the sweep's private counter word, the QUEUED bit of nx_misc; 4: a queue of CW_LA_DEPTH records per env

struct CkptSection { void *dev; size_t bytes; };
static std::vector<CkptSection> ckpt_sections(cw_engine *e)
{
    const CwParams &P = e->P;
    // (file order; the sizes are cw_host.cpp's: cwh_ckpt_section_bytes -- the RNG streams are one reset ahead wherever a look-ahead record waits)
    void *dev[CWH_CKPT_SECTIONS] = {P.hdr, P.pos, P.init_pos, P.goal_pos, P.goal_codes, P.init_agent, P.goal_agent, P.ep_no, P.mt, P.mt_idx, P.pool, P.reward, P.done,
                                    P.achieved_out, P.desired_out, P.episode_length, P.episode_return, P.counters, P.nx_init_pos, P.nx_goal_pos, P.nx_misc,
                                    P.nx_ctl};
    size_t bytes[CWH_CKPT_SECTIONS];
    const int n = cwh_ckpt_section_bytes(e->n, e->K, e->P.lookahead ? CW_LA_DEPTH : 0, bytes, nullptr);
    std::vector<CkptSection> out;
    for (int i = 0; i < n; i++) out.push_back({dev[i], bytes[i]});
    return out;
}
static uint64_t menus_hash(const cw_engine *e)
{
    uint64_t h = 1469598103934665603ull;                 // FNV-1a over the device-form menus
    const unsigned char *b = (const unsigned char *)e->menus.data();
    for (size_t i = 0; i < e->menus.size() * sizeof(CwMenuDev); i++) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}
static CwCkptHeader ckpt_header(cw_engine *e)
{
    CwCkptHeader h{};
    memcpy(h.magic, CW_CKPT_MAGIC, 8);
    h.version = CW_CKPT_VERSION;
    h.header_bytes = (uint32_t)sizeof(CwCkptHeader);
    h.n_envs = e->n; h.size = e->S; h.max_steps = e->P.max_steps; h.pool_k = e->K;
    h.task_mask = e->P.task_mask; h.n_menus = (uint32_t)e->menus.size();
    h.lookahead = (uint32_t)e->P.lookahead;
    h.menus_hash = menus_hash(e);
    h.total_bytes = sizeof(CwCkptHeader);
    for (const CkptSection &sec : ckpt_sections(e)) h.total_bytes += sec.bytes;
    return h;
}

extern "C" {

size_t cw_checkpoint_bytes(cw_engine *e) { return e ? (size_t)ckpt_header(e).total_bytes : 0; }

int cw_checkpoint_save(cw_engine *e, void *buf, size_t capacity)
{
    if (!e || !buf) return fail(CW_ERR_INVALID, "cw_checkpoint_save: null argument");
    if (!e->has_reset) return fail(CW_ERR_STATE, "cw_checkpoint_save called before cw_reset");
    DeviceGuard guard(e->device);
    if (!guard.ok) return fail(CW_ERR_HIP, "hipSetDevice(%d) failed", e->device);
    PARK(e);
    const CwCkptHeader h = ckpt_header(e);
    if (capacity < h.total_bytes) return fail(CW_ERR_INVALID, "cw_checkpoint_save: buffer of %zu bytes, %llu needed", capacity, (unsigned long long)h.total_bytes);
    HIP_TRY(quiesce(e));
    AuxDrain drain(e);                               // (every exit waits for the private stream: the copies below target this frame's buffers)
    unsigned char *p = (unsigned char *)buf;
    memcpy(p, &h, sizeof(h));
    p += sizeof(h);
    for (const CkptSection &sec : ckpt_sections(e)) {
        if (sec.bytes) HIP_TRY(aux_copy(e, p, sec.dev, sec.bytes, hipMemcpyDefault));
        p += sec.bytes;
    }
    HIP_TRY(hipStreamSynchronize(e->aux));
    return CW_OK;
}

int cw_checkpoint_load(cw_engine *e, const void *buf, size_t length)
{
    if (!e || !buf) return fail(CW_ERR_INVALID, "cw_checkpoint_load: null argument");
    DeviceGuard guard(e->device);
    if (!guard.ok) return fail(CW_ERR_HIP, "hipSetDevice(%d) failed", e->device);
    PARK(e);
    CwCkptHeader h;
    if (length < sizeof(h)) return fail(CW_ERR_INVALID, "cw_checkpoint_load: %zu bytes is not a checkpoint", length);
    memcpy(&h, buf, sizeof(h));
    const CwCkptHeader mine = ckpt_header(e);
    if (memcmp(h.magic, CW_CKPT_MAGIC, 8) != 0) return fail(CW_ERR_INVALID, "cw_checkpoint_load: not a CraftingWorld checkpoint");
    if (h.version != CW_CKPT_VERSION || h.header_bytes != sizeof(CwCkptHeader))
        return fail(CW_ERR_INVALID, "cw_checkpoint_load: checkpoint version %u, this library reads version %d", h.version, (int)CW_CKPT_VERSION);
    if (h.n_envs != mine.n_envs || h.size != mine.size || h.max_steps != mine.max_steps || h.pool_k != mine.pool_k ||
        h.task_mask != mine.task_mask)
        return fail(CW_ERR_INVALID, "cw_checkpoint_load: checkpoint of %d envs, size %d, max_steps %d, fixed_init_state %d, task mask %#x; "
                    "this engine: %d, %d, %d, %d, %#x", h.n_envs, h.size, h.max_steps, h.pool_k, h.task_mask, mine.n_envs, mine.size,
                    mine.max_steps, mine.pool_k, mine.task_mask);
    if (h.n_menus != mine.n_menus || h.menus_hash != mine.menus_hash)
        return fail(CW_ERR_INVALID, "cw_checkpoint_load: the checkpoint was written with different task menus (selected_tasks / "
                    "number_of_tasks / stacking / reward_style)");
    // A checkpoint written WITH look-ahead records can be resumed by an engine that keeps none (CW_TUNE_LOOKAHEAD=0, host-mapped outputs) and the
    // other way round: the records are not state, only work done ahead -- where one waits, the env's stream is rewound by the draws it took
    // (cwh_mt_rewind) and the record dropped; an engine that keeps records recomputes them at its next refill.
    const size_t N = (size_t)e->n;
    std::vector<uint32_t> key(CW_MT_N), words(CW_MT_N);      // (scratch of the rewind below; declared ahead of the drain guard: it outlives every copy)
    const size_t la_bytes = N * 16 * 3 * CW_LA_DEPTH + N * 4;
    const unsigned long long expect = mine.total_bytes + (h.lookahead && !mine.lookahead ? la_bytes : 0) - (!h.lookahead && mine.lookahead ? la_bytes : 0);
    if (h.total_bytes != expect || length < h.total_bytes)
        return fail(CW_ERR_INVALID, "cw_checkpoint_load: truncated checkpoint (%zu of %llu bytes)", length, (unsigned long long)h.total_bytes);
    HIP_TRY(quiesce(e));
    AuxDrain drain(e);                               // (every exit waits for the private stream: the copies below target this frame's buffers)
    const unsigned char *p = (const unsigned char *)buf + sizeof(h);
    const unsigned char *blob_mt = nullptr, *blob_idx = nullptr;
    for (const CkptSection &sec : ckpt_sections(e)) {
        if (sec.dev == (void *)e->P.mt) blob_mt = p;
        if (sec.dev == (void *)e->P.mt_idx) blob_idx = p;
        const bool la_section = sec.dev == (void *)e->P.nx_init_pos || sec.dev == (void *)e->P.nx_goal_pos || sec.dev == (void *)e->P.nx_misc || sec.dev == (void *)e->P.nx_ctl;
        if (la_section && h.lookahead != mine.lookahead) continue;           // (handled below)
        if (sec.bytes) HIP_TRY(aux_copy(e, sec.dev, p, sec.bytes, hipMemcpyDefault));
        p += sec.bytes;
    }
    if (h.lookahead && !mine.lookahead) {            // p: the file's look-ahead sections (nx_init_pos, nx_goal_pos, nx_misc, list, count)
        const unsigned char *misc_bytes = p + N * 32 * CW_LA_DEPTH;     // (nx_misc of the file: 4 words per record at an offset that need not be 4-byte aligned -- the 'done' section is N bytes)
        HIP_TRY(hipStreamSynchronize(e->aux));
        for (size_t i = 0; i < N; i++) {
            uint32_t misc[4] = {0, 0, 0, 0};         // [3]: the draws of every record that waited, summed
            for (size_t d = 0; d < CW_LA_DEPTH; d++) {
                uint32_t md[4];
                memcpy(md, misc_bytes + (d * N + i) * 16, 16);
                if (md[2] >> 31) { misc[2] = md[2]; misc[3] += md[3]; }
            }
            if (!(misc[2] >> 31)) continue;
            int32_t pos = 0;
            memcpy(&pos, blob_idx + i * 4, 4);
            memcpy(words.data(), blob_mt + i * CW_MT_N * 4, CW_MT_N * 4);
            cwh_mt_to_numpy(words.data(), pos, key.data());
            cwh_mt_rewind(key.data(), &pos, misc[3]);
            const int32_t idx = cwh_mt_from_numpy(key.data(), pos);
            HIP_TRY(aux_copy(e, e->P.mt + i * CW_MT_N, key.data(), CW_MT_N * 4, hipMemcpyHostToDevice));
            HIP_TRY(aux_copy(e, e->P.mt_idx + i, &idx, 4, hipMemcpyHostToDevice));
            HIP_TRY(hipStreamSynchronize(e->aux));                           // (key / idx are reused by the next env)
        }
    } else if (!h.lookahead && mine.lookahead) {
        HIP_TRY(lookahead_drop(e));                  // no record anywhere: the next refill computes every env's
    }
    e->has_reset = true;
    e->la_refill_all = e->P.lookahead != 0;          // (harmless: envs that hold a record are skipped)
    if (e->obs_mode != CW_OBS_STATE) HIP_TRY(cwk_launch_render_restore(&e->P, &e->tune, e->aux));   // frames follow the records
    HIP_TRY(hipStreamSynchronize(e->aux));
    return CW_OK;
}

}  // extern "C"
