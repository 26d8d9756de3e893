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

extern "C" {
hipError_t cwk_launch_step(const CwParams *P, const CwTuning *T, const void *actions, int act_dtype, int obs_mode, int auto_reset, hipStream_t st,
                           hipStream_t side, hipEvent_t ev_fork, hipEvent_t ev_join, hipEvent_t *ev);
int cwk_render_is_linear(const CwParams *P, const CwTuning *T);
int cwk_render_is_piece_sweep(const CwParams *P, const CwTuning *T);
hipError_t cwk_launch_reset_all(const CwParams *P, const CwTuning *T, int obs_mode, hipStream_t st);
hipError_t cwk_launch_pool(const CwParams *P, const CwTuning *T, hipStream_t st);
hipError_t cwk_launch_seed(const CwParams *P, const uint32_t *seeds_dev, hipStream_t st);
hipError_t cwk_launch_resident(const CwParams *P, CwResident *R, uint32_t seq0, int paint_dirty, unsigned long long idle_ticks,
                               unsigned long long life_ticks, hipStream_t st);
hipError_t cwk_launch_render_onehot(const CwParams *P, const uint8_t *onehot, int n_states, uint16_t *out, hipStream_t st);
hipError_t cwk_launch_render_restore(const CwParams *P, const CwTuning *T, hipStream_t st);
hipError_t cwk_launch_rollout(const CwParams *P, const uint8_t *actions, int T, int32_t *rewards, uint8_t *dones, hipStream_t st);
hipError_t cwk_launch_render_ext(const CwParams *P, const CwTuning *T, uint8_t *out, hipStream_t st);
hipError_t cwk_launch_step_render_calib(const CwParams *P, const CwTuning *T, int auto_reset, hipStream_t st);
hipError_t cwk_launch_render_calib(const CwParams *P, const CwTuning *T, hipStream_t st, int q_all, int fast_parity, int *blocks,
                                   int *waves_per_block);
int cwk_render_jobs(const CwParams *P, const CwTuning *T);
int cwk_step_renders_fused(const CwParams *P, const CwTuning *T, int auto_reset);
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

enum { CW_ADAPT_RING = 256 };
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
    hipStream_t last_stream = nullptr; // the stream of the last cw_reset / cw_step / cw_rollout: a resident kernel starts only after that work
    bool last_stream_set = false;
    std::vector<hipEvent_t> prof_ev;   // 6 per recorded step
    hipStream_t side = nullptr;        // reset + reset-render run here beside the main render (FULL pixel mode)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int prof_cap = 0, prof_n = 0;
    // online tuner of the render pace (full-frame mode, linear sweep): see adapt_tick
    struct Adapt {
        bool on = false;
        hipEvent_t ev[CW_ADAPT_RING] = {nullptr};   // ev[w % RING] is recorded on the caller's stream when window w begins (the host may run
                                               // RING windows = 2 048 steps ahead of the GPU before measurements are lost)
        unsigned seq = 0;                      // steps taken in the tuned mode
        unsigned next_window = 0;              // first window whose duration has not been read yet
        int cur = 2;                           // extra sleeps per pair of jobs while envs are reset beside the sweep, currently held
        signed char pace_of_window[CW_ADAPT_RING] = {0};  // what each recent window ran at; negative: a settling window, not counted
        float stat[16] = {0};                  // per pace: running mean step time of its counted windows (ms; 0: unknown)
        unsigned stat_window[16] = {0};        // window of the newest sample in stat[]
        // ... and of the PLACEMENT of the sweep's batch loop (cw_render_step_kernel<k>, cw_kernels.hip: render_groups)
        bool pace_on = false;                  // (1) is tuned (off: CW_TUNE_RENDER_PACE_BESIDE forces the number)
        bool place_on = false;                 // (2) is tuned (off: CW_TUNE_RENDER_PLACE forces one)
        int place = 3;                         // the placement held outside a survey
        bool surveying = false;
        unsigned survey_w0 = 0;                // first window of the running survey
        signed char place_of_window[CW_ADAPT_RING] = {0}; // placement each recent window ran at
        signed char round_of_window[CW_ADAPT_RING] = {0}; // 0: not a survey window; r + 1: round r of a survey (round 0 is not counted)
        bool tainted[CW_ADAPT_RING] = {false};            // a step of the window was bracketed by cw_profile_* events (each costs a pipeline bubble): not counted
        float survey_ms[8][3] = {{0}};         // ms per step of placement k in survey rounds 1..3
        unsigned survey_seen = 0;              // survey windows read so far
        float place_ms = 0;                    // what `place` measured when it was chosen
        int place_bad = 0;                     // consecutive counted windows more than 4 % above that
        unsigned surveys = 0;
        unsigned place_struck = 0;             // bit k: placement k was held and fell out of its regime (not held again in this process)
        bool guard_on = false;                 // (3) the piece sweep runs unpaced and its regime is watched (cwh_regime_guard)
        float guard[41] = {};
        int guard_pace = 0;                    //     what the guard last said: 0 unpaced, 1 paced
        bool guard_test = false;
        int guard_trials = 0, gpace_last = 0;
        unsigned gsurvey_w0 = 16;
        signed char gpace_of_window[CW_ADAPT_RING] = {};   // what the window was launched with: 0 unpaced, 1 paced, -1 the first after a change (not counted)
    } adapt;
};
enum { CW_ADAPT_W = 8, CW_ADAPT_MAX = 8, CW_PLACES = 8, CW_SURVEY_ROUNDS = 4, CW_PLACE_BAD_WINDOWS = 24, CW_GUARD_PACE = 2 /* eighths */ };
extern "C" int cwh_regime_guard(float *s, float ms, unsigned window, int ran_paced);

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

// ------------------------------------------------------------------------------ MT19937 (host)
// numpy RandomState (key, pos)  <->  the engine's consume-and-replace form (cw_mt.h).
static inline uint32_t mt_twist(uint32_t cur, uint32_t nxt, uint32_t far)
{
    const uint32_t y = (cur & 0x80000000u) | (nxt & 0x7fffffffu);
    return far ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

extern "C" {

// in place: words < pos become next-generation (numpy's twist loop, first `pos` iterations);
// returns the engine index (pos mod 624)
int cwh_mt_from_numpy(uint32_t *s, int pos)
{
    if (pos < 0) pos = 0;
    if (pos > CW_MT_N) pos = CW_MT_N;
    for (int k = 0; k < pos; k++)
        s[k] = mt_twist(s[k], s[(k + 1) % CW_MT_N], s[(k + 397) % CW_MT_N]);
    return pos % CW_MT_N;
}

// inverse: from engine form (s, idx) recover a numpy key whose stream from position idx is
// identical.  Words < idx are un-twisted; key[0]'s low 31 bits are unrecoverable and unused
// by MT19937 (set to 0).
void cwh_mt_to_numpy(const uint32_t *s, int idx, uint32_t *key)
{
    for (int j = idx; j < CW_MT_N; j++) key[j] = s[j];
    for (int j = 0; j < idx; j++) key[j] = 0;
    for (int j = idx - 1; j >= 0; j--) {
        const uint32_t m = (j < CW_MT_N - 397) ? key[j + 397] : s[j - (CW_MT_N - 397)];
        uint32_t t = s[j] ^ m;
        const uint32_t odd = t >> 31;
        if (odd) t ^= 0x9908b0dfu;
        const uint32_t y = (t << 1) | odd;     // (G[j] & UPPER) | (G[j+1] & LOWER)
        key[j] |= y & 0x80000000u;
        if (j + 1 < idx) key[j + 1] |= y & 0x7fffffffu;
    }
}

// ------------------------------------------------------------------------------ DLPack producer
// Non-owning DLManagedTensor over engine memory (DLPack ABI v0: the struct layout below is the
// published one).  Produced and freed in C so that no Python callback is involved when a consumer
// (torch) drops its last view -- possibly during interpreter shutdown.
struct CwDLDevice { int32_t device_type, device_id; };
struct CwDLDataType { uint8_t code, bits; uint16_t lanes; };
struct CwDLTensor { void *data; CwDLDevice device; int32_t ndim; CwDLDataType dtype; int64_t *shape, *strides; uint64_t byte_offset; };
struct CwDLManagedTensor { CwDLTensor dl_tensor; void *manager_ctx; void (*deleter)(CwDLManagedTensor *); };

static void cw_dl_deleter(CwDLManagedTensor *m)
{
    if (!m) return;
    free(m->dl_tensor.shape);
    free(m);
}

// device_type 10 = kDLROCM; code 0 int / 1 uint; returns a malloc'ed DLManagedTensor* (or NULL)
void *cwh_dlpack_make(void *data, int device_id, int code, int bits, int ndim, const int64_t *shape)
{
    CwDLManagedTensor *m = (CwDLManagedTensor *)calloc(1, sizeof(CwDLManagedTensor));
    int64_t *shp = (int64_t *)malloc(sizeof(int64_t) * (size_t)(ndim > 0 ? ndim : 1));
    if (!m || !shp) { free(m); free(shp); return nullptr; }
    for (int i = 0; i < ndim; i++) shp[i] = shape[i];
    m->dl_tensor.data = data;
    m->dl_tensor.device = CwDLDevice{10, device_id};
    m->dl_tensor.ndim = ndim;
    m->dl_tensor.dtype = CwDLDataType{(uint8_t)code, (uint8_t)bits, 1};
    m->dl_tensor.shape = shp;
    m->dl_tensor.strides = nullptr;
    m->dl_tensor.byte_offset = 0;
    m->deleter = cw_dl_deleter;
    return m;
}

// The placement survey's decision (adapt_tick), as a pure function: med[k] = median ms/step of placement k (<= 0: no figure), struck = bit
// mask of placements not to hold again.  Of the placements within 6 % of the fastest (off the cliff) the MEDIAN one -- the fastest are
// bistable (profiles/r03_placement.txt C).  -> the placement, or -1 if none has a figure; *n_candidates for the log.
int cwh_choose_place(const float *med, unsigned struck, int *n_candidates)
{
    float fastest = 0.f;
    for (int k = 0; k < 8; k++)
        if (med[k] > 0.f && !((struck >> k) & 1u) && (fastest == 0.f || med[k] < fastest)) fastest = med[k];
    int cand[8], n = 0;
    for (int k = 0; k < 8; k++)
        if (med[k] > 0.f && !((struck >> k) & 1u) && med[k] <= 1.06f * fastest) cand[n++] = k;
    std::sort(cand, cand + n, [&](int x, int y) { return med[x] < med[y] || (med[x] == med[y] && x < y); });
    if (n_candidates) *n_candidates = n;
    return n > 0 ? cand[(n - 1) / 2] : -1;
}

// The piece sweep's REGIME GUARD (adapt_tick), as a pure function fed one counted window at a time: its level (ms per step) and the pace it was LAUNCHED
// with (0 unpaced, 1 paced; the host may be many windows ahead of the GPU, so what a window ran with is recorded when it starts, not inferred).
// The unpaced sweep sits just short of the write path's slower, saturated regime (profiles/r03_pieces.txt B, F, N-P); should a process find itself in
// it -- another build, another box, another driver, a neighbour on the memory system -- a paced sweep is the way out: 3 % slower than the good regime,
// 13 % faster than the bad.
//  * It opens with a SURVEY (state 3; the caller starts it with s[2] = 3, everything else 0): six unpaced windows, then six paced ones; paced 3 % faster than
//    unpaced: it stays for the process (-> 2).  (Launches timed at cw_create, without the step kernel in between, do not show a build whose unpaced
//    sweep is in the slower regime from its first step; and later there would be no better level to compare with.)
//  * Watching (0): unpaced windows more than 10 % above the best level seen, 32 in a row -> TRIAL (1) of the paced sweep, 32 paced windows (the first
//    two settle); 3 % faster than the 32 unpaced ones before it: it stays (-> 2); otherwise back to unpaced, the level of those windows is the new normal
//    (the workload changed -- episode phases that spread out cost 12 % --, not the regime), and the next trial has to wait twice as long.
// state: [0] best level, [1] bad windows in a row, [2] the state, [3] paced windows seen (trial) / unpaced (survey), [4] sum of them, [5] mean of the last
// 32 unpaced windows before the trial / paced windows seen (survey), [6] window of the earliest next trial / sum of the paced ones (survey), [7] hold-off,
// [8..39] ring of the last 32 unpaced levels, [40] ring position.
// -> the pace to launch with from now on: 0 unpaced, 1 paced.
int cwh_regime_guard(float *s, float ms, unsigned window, int ran_paced)
{
    enum { BEST, BAD, STATE, TRIAL_N, TRIAL_SUM, BEFORE, NEXT, HOLD, RING = 8, POS = 40 };
    if (s[STATE] == 2.f) return 1;
    if (s[STATE] == 3.f) {
        if (ran_paced) { s[BEFORE] += 1.f; s[NEXT] += ms; if (ms > s[RING + 1]) s[RING + 1] = ms; }
        else { s[TRIAL_N] += 1.f; s[TRIAL_SUM] += ms; if (ms > s[RING]) s[RING] = ms; }
        const int n0 = (int)s[TRIAL_N], n1 = (int)s[BEFORE];
        if (n0 < 6) return 0;
        if (n1 < 6) return 1;
        // (means without each side's slowest window: one may hold a step on which every env was reset)
        const float unpaced_ms = (s[TRIAL_SUM] - s[RING]) / (float)(n0 - 1), paced_ms = (s[NEXT] - s[RING + 1]) / (float)(n1 - 1);
        s[TRIAL_N] = s[TRIAL_SUM] = s[BEFORE] = s[NEXT] = s[RING] = s[RING + 1] = 0.f;
        if (paced_ms < 0.97f * unpaced_ms) { s[STATE] = 2.f; s[BEST] = paced_ms; return 1; }
        s[STATE] = 0.f;
        s[BEST] = unpaced_ms;
        return 0;
    }
    if (s[STATE] == 1.f) {
        if (!ran_paced) return 1;                                          // (launched before the trial began)
        s[TRIAL_N] += 1.f;
        if (s[TRIAL_N] > 2.f) s[TRIAL_SUM] += ms;
        if (s[TRIAL_N] < 32.f) return 1;
        const float trial = s[TRIAL_SUM] / 30.f;
        if (trial < 0.97f * s[BEFORE]) { s[STATE] = 2.f; return 1; }
        s[STATE] = 0.f;                                                    // no: the new normal
        s[BEST] = s[BEFORE];
        s[BAD] = 0.f;
        s[HOLD] = s[HOLD] > 0.f ? 2.f * s[HOLD] : 256.f;
        s[NEXT] = (float)window + s[HOLD];
        return 0;
    }
    if (ran_paced) return 0;                                               // (a straggler of a trial that has ended)
    const int pos = (int)s[POS];
    s[RING + pos] = ms;
    s[POS] = (float)((pos + 1) & 31);
    if (s[BEST] == 0.f || ms < s[BEST]) s[BEST] = ms;
    s[BAD] = ms > 1.10f * s[BEST] ? s[BAD] + 1.f : 0.f;
    if (s[BAD] >= 32.f && (float)window >= s[NEXT]) {
        float sum = 0.f;
        for (int i = 0; i < 32; i++) sum += s[RING + i];
        s[BEFORE] = sum / 32.f;
        s[STATE] = 1.f;
        s[TRIAL_N] = 0.f;
        s[TRIAL_SUM] = 0.f;
        return 1;
    }
    return 0;
}

void cwh_mt_init_genrand(uint32_t *s, uint32_t seed)   // numpy RandomState(int): init_genrand
{
    s[0] = seed;
    for (int i = 1; i < CW_MT_N; i++) s[i] = 1812433253u * (s[i - 1] ^ (s[i - 1] >> 30)) + (uint32_t)i;
}

}  // extern "C"

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

// Pace of the linear-sweep render (cw_kernels.hip: render_groups): idle clocks per pair of jobs.  The write path is less
// efficient saturated than kept just short of saturation.
//
// Ray raster: the pace is FIXED at m+0 (one s_sleep inside every job, none between jobs); on top of it come extra sleeps per pair of
// jobs while at least CW_BESIDE_MIN envs are being reset beside the sweep (cw_kernels.hip: render_groups), and THAT number is what
// the online tuner (adapt_tick) follows.  Round 2 first measured the pace at cw_create (median launch time per candidate) and let
// the tuner follow it; then both were compared with forced paces, alternating on one box, several boxes (profiles/history/r02_pace.txt):
// the launch times cw_create can measure scatter by 4-5 % between processes for the SAME pace (0.2298-0.2455 ms for m+0) -- more
// than the differences to be resolved -- so the calibration picked m+2 / m+3 in a third of the processes.  Inside a step sequence
// the order is the same on every box and shape tried: m+0 0.2333, one sleep per pair without the inner one 0.2332-0.2345, unpaced
// 0.2350, m+1 0.2365, m+2 0.2418 ms (one-launch step, no resets beside; 131 072 envs, 262 144 envs and 32x32 likewise).  With
// resets beside every launch (episode phases spread out) the best extra is box-dependent -- 0 to 3 more sleeps per pair, and being
// off by two costs 3-8 % -- hence the tuner there.  CW_TUNE_RENDER_CALIBRATE=1 brings the cw_create measurement back (other
// hardware), CW_TUNE_RENDER_PACE=n forces the base pace, CW_TUNE_RENDER_PACE_BESIDE=n the extra (tuner off), CW_TUNE_RENDER_ADAPT=0
// keeps the extra at its start value 2.
// AltObs raster: measured here as before (0..6 sleeps per 1-KiB store; the optimum is flat and broad there).
static int calibrate_render_pace(cw_engine *e, bool refine)
{
    CwTuning &tn = e->tune;
    const char *forced = getenv("CW_TUNE_RENDER_PACE");
    if (forced) {                                         // (256 + n: with the sleep inside each job)
        const char *beside = getenv("CW_TUNE_RENDER_PACE_BESIDE");
        tn.render_pace = (atoi(forced) < 0 ? 0 : atoi(forced) & 0x1FF) | (((beside ? atoi(beside) : 2) & 15) << 12);
        if (e->P.raster == CW_RASTER_ALT) e->P.alt_pace = tn.render_pace & 0xFF;
        return CW_OK;
    }
    const bool alt = e->P.raster == CW_RASTER_ALT;       // the AltObs painter's pace lives in P.alt_pace (every kernel that paints frames reads it)
    if (alt) e->P.alt_pace = 2;                           // (engines that are not calibrated: mid-range)
    if (e->obs_mode != CW_OBS_PIXELS_FULL || e->host_actions) return CW_OK;      // (the Ray raster is paced in both of its kernels)
    if ((long long)e->n * e->P.frame_bytes < (64ll << 20)) return CW_OK;          // small batches are launch-bound: nothing to pace
#ifdef CW_EXPERIMENT
    const bool measure_base_pace = getenv("CW_TUNE_RENDER_CALIBRATE") && atoi(getenv("CW_TUNE_RENDER_CALIBRATE")) != 0;
#else
    const bool measure_base_pace = false;
#endif
    if (!alt && !measure_base_pace) {
        const char *beside = getenv("CW_TUNE_RENDER_PACE_BESIDE");
        tn.render_pace = 0x100 | (((beside ? atoi(beside) : 2) & 15) << 12);
        return CW_OK;
    }
    // The launches of one candidate are queued back to back and the host waits once, at the end: a host round trip after every
    // launch lets the card idle for tens of microseconds each time, and what is measured then is a memory system that keeps
    // leaving and re-entering its busy state (the same kernel on the same buffer reads 0.225 or 0.28 ms that way,
    // tools/microbench/time_render.py placement) -- not the regime a step sequence runs in.
    enum { CALIB_LAUNCHES = 9, CALIB_SKIP = 3 };
    hipEvent_t evs[2 * CALIB_LAUNCHES] = {};
    for (hipEvent_t &ev : evs)
        if (hipEventCreate(&ev) != hipSuccess) {
            for (hipEvent_t &d : evs) if (d) (void)hipEventDestroy(d);
            return fail(CW_ERR_HIP, "cw_create: event creation failed");
        }
    int blocks = 0, wpb = 0, rc = CW_OK;
    auto median_ms = [&](int pace, double *out) -> int {
        const int saved = tn.render_pace, saved_alt = e->P.alt_pace;
        tn.render_pace = pace;
        if (alt) e->P.alt_pace = pace;
        bool ok = true;
        for (int rep = 0; rep < CALIB_LAUNCHES && ok; rep++)             // (a short idle kernel between launches, like the step kernel)
            ok = hipEventRecord(evs[2 * rep], nullptr) == hipSuccess &&
                 cwk_launch_render_calib(&e->P, &tn, nullptr, tn.render_q_all, tn.render_fast_parity, &blocks, &wpb) == hipSuccess &&
                 hipEventRecord(evs[2 * rep + 1], nullptr) == hipSuccess && cwk_launch_idle(nullptr) == hipSuccess;
        ok = ok && hipDeviceSynchronize() == hipSuccess;
        tn.render_pace = saved;
        e->P.alt_pace = saved_alt;
        float ms[CALIB_LAUNCHES];
        for (int rep = 0; rep < CALIB_LAUNCHES && ok; rep++) ok = hipEventElapsedTime(&ms[rep], evs[2 * rep], evs[2 * rep + 1]) == hipSuccess;
        if (!ok) return fail(CW_ERR_HIP, "cw_create: render pace calibration failed");
        float *m = ms + CALIB_SKIP;                                             // (the first launches bring the card up to speed)
        const int n = CALIB_LAUNCHES - CALIB_SKIP;
        for (int i = 1; i < n; i++) for (int j = i; j > 0 && m[j] < m[j - 1]; j--) { const float t = m[j]; m[j] = m[j - 1]; m[j - 1] = t; }
        *out = m[n / 2];
        return CW_OK;
    };
    int best = tn.render_pace;
    double best_ms = 0, t = 0;
    char log[400] = "";
    size_t len = 0;
    if (!refine) for (int i = 0; i < 3 && rc == CW_OK; i++) rc = median_ms(alt ? 2 : 0x100, &t);   // the card up to speed before the first candidate is timed
    auto try_pace = [&](int pace) {
        rc = median_ms(pace, &t);
        if (rc == CW_OK && (best_ms == 0 || t < best_ms)) { best_ms = t; best = pace; }
        if (len < sizeof(log) - 24) len += (size_t)snprintf(log + len, sizeof(log) - len, " %s%d:%.4f", (pace & 0x100) ? "m+" : "", pace & 0xFF, t);
    };
    if (alt) {
        if (!refine) for (int pp = 0; pp <= 6 && rc == CW_OK; pp++) try_pace(pp);
        else best = e->P.alt_pace;
    } else if (!refine) {
        static const int cand[] = {0x100, 0x101, 0x102, 0x103, 0x104, 0, 1, 2, 3, 4, 6};
        for (size_t i = 0; i < sizeof(cand) / sizeof(cand[0]) && rc == CW_OK; i++) try_pace(cand[i]);
    } else {                                                                    // after the shares are known: the neighbours once more
        const int mid = tn.render_pace & 0x100, p0 = tn.render_pace & 0xFF;      // (bits 12-15, the extra beside resets, play no part here)
        for (int pp = (p0 > 0 ? p0 - 1 : 0); pp <= p0 + 1 && rc == CW_OK; pp++) try_pace(mid | pp);
    }
    for (hipEvent_t &ev : evs) (void)hipEventDestroy(ev);
    if (rc != CW_OK) return rc;
    if (alt) e->P.alt_pace = best;
    else tn.render_pace = (best & 0x1FF) | (2 << 12);       // (+2 while envs are reset beside the launch, see render_groups)
    if (getenv("CW_TUNE_VERBOSE"))
        fprintf(stderr, "[craftingworld] render pace%s: ms per launch by sleeps per pair of jobs (m+: and one inside each job)%s -> %s%d\n",
                refine ? " (with shares)" : "", log, (best & 0x100) ? "m+" : "", best & 0xFF);
    return CW_OK;
}

// launch time of the per-step render as currently configured (launches queued back to back, one wait: see calibrate_render_pace): the median and the
// 90th percentile of 20 launches -- the write path has a slower regime that a configuration may enter launch by launch (profiles/r03_pieces.txt N-P), and
// a median does not show a configuration that does so one time in three
static int timed_render_stats(cw_engine *e, double *median, double *p90)
{
    enum { LAUNCHES = 24, SKIP = 4 };
    hipEvent_t evs[2 * LAUNCHES] = {};
    bool ok = true;
    for (hipEvent_t &ev : evs) ok = ok && hipEventCreate(&ev) == hipSuccess;
    for (int rep = 0; rep < LAUNCHES && ok; rep++)
        ok = hipEventRecord(evs[2 * rep], nullptr) == hipSuccess && cwk_launch_step_render_calib(&e->P, &e->tune, e->auto_reset, nullptr) == hipSuccess &&
             hipEventRecord(evs[2 * rep + 1], nullptr) == hipSuccess && cwk_launch_idle(nullptr) == hipSuccess;
    ok = ok && hipDeviceSynchronize() == hipSuccess;
    float ms[LAUNCHES];
    for (int rep = 0; rep < LAUNCHES && ok; rep++) ok = hipEventElapsedTime(&ms[rep], evs[2 * rep], evs[2 * rep + 1]) == hipSuccess;
    for (hipEvent_t &ev : evs) if (ev) (void)hipEventDestroy(ev);
    if (!ok) return fail(CW_ERR_HIP, "cw_create: render calibration failed");
    std::sort(ms + SKIP, ms + LAUNCHES);
    const int n = LAUNCHES - SKIP;
    *median = ms[SKIP + n / 2];
    *p90 = ms[SKIP + (9 * n) / 10 - 1];
    return CW_OK;
}

// The per-step render: the painter calibrated so far (the Ray raster's linear sweep of cell rows, or frame per wave) or the sweep of aligned
// 4-KiB pieces (cw_kernels.hip: render_pieces)?  The latter's pace (eighths of a sleep per 1-KiB store) is checked, then the faster of the two
// is kept -- the same launches, queued the same way, for both.
static int calibrate_piece_sweep(cw_engine *e)
{
    CwTuning &tn = e->tune;
    if (!tn.piece_sweep || !cwk_render_is_piece_sweep(&e->P, &tn)) return CW_OK;
    tn.piece_pace = e->P.raster == CW_RASTER_ALT ? 4 : 0;
    // the extra QUARTER sleeps per store while envs are being reset beside the sweep: bits 12-15 of render_pace, the number cw_step's tuner follows
    // from here (episode phases spread out, 65 536 envs: 0 / 2 / 4 / 6 / 8 / 12 extra = 0.250-0.260 / 0.236-0.257 / 0.238-0.248 / 0.234-0.237 /
    // 0.241 / 0.241-0.248 ms, profiles/r03_pieces.txt; AltObs: 4)
    auto set_beside = [&]() {
        const char *beside = getenv("CW_TUNE_RENDER_PACE_BESIDE");
        tn.render_pace = (tn.render_pace & ~0xF000) | (((beside ? atoi(beside) : e->P.raster == CW_RASTER_ALT ? 4 : 6) & 15) << 12);
    };
#ifdef CW_EXPERIMENT
    const char *forced = getenv("CW_TUNE_PIECE_PACE");
#else
    const char *forced = nullptr;
#endif
    if (forced) { tn.piece_pace = atoi(forced) < 0 ? 0 : atoi(forced) & 0xFF; set_beside(); return CW_OK; }
    // (batches down to 4 MB of frames are measured: which painter wins a launch-bound render depends on the shape -- 700 envs of 70x70: frame per wave
    // 0.028 ms, pieces 0.044)
    if (e->obs_mode != CW_OBS_PIXELS_FULL || e->host_actions || (long long)e->n * e->P.frame_bytes < (4ll << 20)) { set_beside(); return CW_OK; }
    double frames_ms = 0, frames_p90 = 0, t = 0, t90 = 0, best_p90 = 0;
    tn.piece_sweep = 0;
    int rc = timed_render_stats(e, &frames_ms, &frames_p90);
    tn.piece_sweep = 1;
    // Which pace.  The sweep runs fastest unpaced or nearly so -- just short of the write path's slower, saturated regime -- and how much pace it takes to
    // stay clear of that regime depends on details of the build (the committed one: unpaced 0.2107-0.2121 ms in every process measured; a build two
    // instructions per batch heavier: unpaced 0.229-0.244, one eighth bimodal 0.210 / 0.239, two eighths a steady 0.2089; profiles/r03_pieces.txt F, P).
    // So the candidates are judged by their 90th-percentile launch, not their median: the smallest pace whose slow launches are within 1.5 % of the best
    // candidate's.  (AltObs: 4 eighths first, a sleep after every other store -- 0.1304-0.1322 ms on four boxes, profiles/r03_alt_sweep.txt)
    static const int eighths_ray[] = {0, 1, 2, 4, 8}, eighths_alt[] = {4, 0, 2, 8, 12};
    const int *eighths = e->P.raster == CW_RASTER_ALT ? eighths_alt : eighths_ray;
    double p90s[5] = {0, 0, 0, 0, 0};
    char log[320] = "";
    size_t len = 0;
    for (size_t i = 0; i < 5 && rc == CW_OK; i++) {
        tn.piece_pace = eighths[i];
        rc = timed_render_stats(e, &t, &t90);
        p90s[i] = t90;
        if (rc == CW_OK && (best_p90 == 0 || t90 < best_p90)) best_p90 = t90;
        if (len < sizeof(log) - 24) len += (size_t)snprintf(log + len, sizeof(log) - len, " %d:%.4f/%.4f", eighths[i], t, t90);
    }
    if (rc != CW_OK) return rc;
    int best = eighths[0];
    for (size_t i = 0; i < 5; i++)
        if (p90s[i] <= 1.015 * best_p90) { best = eighths[i]; break; }             // (the first in the list's order of preference)
    tn.piece_pace = best;
    if (best_p90 >= frames_p90) tn.piece_sweep = 0;
    if (tn.piece_sweep) set_beside();
    if (getenv("CW_TUNE_VERBOSE"))
        fprintf(stderr, "[craftingworld] per-step render: %s %.4f/%.4f ms (median/90th percentile of 20 launches); sweep of aligned pieces by eighths of a sleep per store%s -> %s%s\n",
                cwk_render_is_linear(&e->P, &tn) ? "sweep of cell rows" : "frame per wave", frames_ms, frames_p90, log, tn.piece_sweep ? "pieces, pace " : "the former",
                tn.piece_sweep ? std::to_string(tn.piece_pace).c_str() : "");
    return CW_OK;
}

// Online tuner of the one-launch full-frame step: (1) the sweep's extra sleeps beside resets, (2) the placement of its batch loop.
// Neither can be predicted from launches timed at cw_create (profiles/history/r02_pace.txt, r02_fused_render.txt, r03_placement.txt), so cw_step
// keeps measuring the thing itself: an event is recorded on the caller's stream every CW_ADAPT_W steps (a "window"), and the time between
// two consecutive ones, read whenever both have completed -- however far the host runs ahead of the GPU -- is what CW_ADAPT_W whole steps
// took.  Only performance depends on any of it: every placement and every pace paints the same frames.
// (1) Windows follow a fixed cycle of 24: twenty at `cur`, two at cur + 1, two at cur - 1 (the first window after a change settles and is
// not counted; a window holding a step on which every env was reset is an outlier and is not counted either); each counted window updates
// the running figure of its value, and `cur` moves to a neighbour whose figure is 0.7 % better (figures older than three cycles do not
// count).  With fewer than CW_BESIDE_MIN resets per step the value is never used by the kernel and its drift is harmless.
// (2) The same instructions run up to 17 % apart depending on where the batch loop lies modulo 32 bytes, and which placement is the good one
// changes with the loop body, the compiler and the box.  So all eight are built (cw_render_step_kernel<k>) and a SURVEY picks one: four
// rounds of one window per placement (the first round -- a variant's first launches load its code -- is not counted); of the placements
// whose median window is within 6 % of the fastest, the MEDIAN one is then held (the fastest are bistable, see below).  A survey runs when
// the engine starts stepping and again when the held placement has read more than 4 % above its own survey figure for
// CW_PLACE_BAD_WINDOWS counted windows in a row (the regime has moved: e.g. episode phases that have spread out); that placement is
// then struck off for the rest of the process.  The pace of (1) is frozen during a survey.  CW_TUNE_RENDER_PLACE=k forces a placement.
static void adapt_tick(cw_engine *e, hipStream_t st)
{
    cw_engine::Adapt &a = e->adapt;
    const unsigned w = a.seq / CW_ADAPT_W;           // the window about to start
    if (hipEventRecord(a.ev[w % CW_ADAPT_RING], st) != hipSuccess) return;
    a.tainted[w % CW_ADAPT_RING] = false;
    const bool verbose = getenv("CW_TUNE_VERBOSE") != nullptr;
    bool moved = false;
    while (a.next_window + 1 <= w) {                 // window next_window lies between ev[next_window] and ev[next_window + 1]
        const unsigned cw = a.next_window;
        if (w - cw >= CW_ADAPT_RING - 1) {                                                      // (its events have been reused)
            if (a.round_of_window[cw % CW_ADAPT_RING]) a.survey_seen++;                     //  a survey window lost: its sample stays 0 = unknown
            a.next_window++;
            continue;
        }
        if (cw + 1 == w) break;                                                  // its closing event was recorded just now
        if (hipEventQuery(a.ev[(cw + 1) % CW_ADAPT_RING]) != hipSuccess) break;
        float ms = 0.f;
        const int p = a.pace_of_window[cw % CW_ADAPT_RING];
        const int round = a.round_of_window[cw % CW_ADAPT_RING], place = a.place_of_window[cw % CW_ADAPT_RING];
        a.next_window++;
        const bool timed = !a.tainted[cw % CW_ADAPT_RING] && hipEventElapsedTime(&ms, a.ev[cw % CW_ADAPT_RING], a.ev[(cw + 1) % CW_ADAPT_RING]) == hipSuccess && ms > 0.f;
        ms /= (float)CW_ADAPT_W;
        if (round) {                                                             // a survey window
            if (timed && round >= 2) a.survey_ms[place & 7][round - 2] = ms;
            a.survey_seen++;
            continue;
        }
        if (a.guard_on && timed) {                                            // (3) the piece sweep's regime: every timed window, whatever the extra pace beside resets
            // (guard_test, experiment build: every watched window reads 20 % above the best level, so trials come round by themselves -- for the test
            // that a trial changes no frame)
            const int ran = a.gpace_of_window[cw % CW_ADAPT_RING];
            if (ran >= 0) {                                                       // (else: the first window after a change of the pace settles)
                const float state_before = a.guard[2];
                const int want = cwh_regime_guard(a.guard, a.guard_test && a.guard[2] == 0.f && a.guard[0] > 0.f ? 1.2f * a.guard[0] : ms, cw, ran);
                if (verbose && state_before == 3.f && a.guard[2] != 3.f)
                    fprintf(stderr, "[craftingworld] regime guard, opening survey (window %u): %s (level %.4f ms/step)\n", cw,
                            a.guard[2] == 2.f ? "the paced sweep is 3 % faster than the unpaced one: it stays" : "the unpaced sweep stays", a.guard[0]);
                if (a.guard[2] == 3.f) {                                              // (the opening survey runs on its schedule)
                } else if (want != a.guard_pace) {
                    if (verbose && a.guard[2] != 3.f)
                        fprintf(stderr, "[craftingworld] regime guard (window %u): %.4f ms/step, best level %.4f -> %s\n", cw, ms, a.guard[0],
                                want ? (a.guard[2] == 2.f ? "the paced sweep stays" : "trying the paced sweep") : "the unpaced sweep (the paced one is not 3 % faster)");
                    a.guard_pace = want;
                    if (want && state_before != 3.f) a.guard_trials++;
                } else if (verbose && want && a.guard[2] == 2.f && a.guard[3] == 32.f) {
                    fprintf(stderr, "[craftingworld] regime guard (window %u): the paced sweep is 3 %% faster than the 32 windows before it: it stays\n", cw);
                    a.guard[3] = 33.f;
                }
            }
        }
        if (p < 0 || !timed) continue;
        // is the held placement still what it was?  (CONSECUTIVE windows: one holding an all-env reset step is followed by a normal one;
        // counted before the outlier filter below, which would take a placement that has tipped by 17 % for a reset storm 80 windows long)
        if (a.place_on && !a.surveying && a.place_ms > 0 && p == a.cur) a.place_bad = ms > 1.04f * a.place_ms ? a.place_bad + 1 : 0;
        const bool known = a.stat[p] > 0 && cw - a.stat_window[p] < 80;
        if (known && ms > 1.06f * a.stat[p]) continue;                           // a reset storm inside the window
        a.stat[p] = known ? 0.5f * (a.stat[p] + ms) : ms;
        a.stat_window[p] = cw;
        moved = true;
    }
    if (a.guard_on) {                                // what window w is launched with (the first window after a change settles and is not counted)
        int gp = a.guard_pace;
        bool skip = false;
        if (a.guard[2] == 3.f) {
            // the opening survey is SCHEDULED by window number, not steered by what has been read: a host that enqueues ahead (bench.py: hundreds of
            // windows) gets the guard's answers that much later.  After the first 16 windows (a card that idled through set-up runs its first
            // launches a few per cent slower): 4 unpaced, 4 paced, 4 unpaced, 4 paced (the first of each four settles), unpaced from there on until
            // all of them have been read and the guard has decided (repeated if the windows were lost to the ring).
            if (w >= a.gsurvey_w0 && w - a.gsurvey_w0 >= CW_ADAPT_RING + 16) a.gsurvey_w0 = w;
            const unsigned j = w - a.gsurvey_w0;
            gp = (w >= a.gsurvey_w0 && j < 16) ? (int)((j >> 2) & 1) : 0;
            skip = w < a.gsurvey_w0 || (j < 16 && (j & 3) == 0) || j == 16;
        }
        a.gpace_of_window[w % CW_ADAPT_RING] = (signed char)(gp != a.gpace_last || skip ? -1 : gp);
        a.gpace_last = gp;
        e->tune.piece_pace = gp ? CW_GUARD_PACE : 0;
    }
    if (a.surveying && a.survey_seen >= (unsigned)(CW_SURVEY_ROUNDS * CW_PLACES) && w >= a.survey_w0 + CW_SURVEY_ROUNDS * CW_PLACES) {
        // Every survey window has been read.  WHICH placement to hold: not the fastest.  Placements come in three kinds (profiles/
        // r03_placement.txt): on the cliff (+15-25 %), on the plateau (within ~1 % of each other), and one or two 2-3 % FASTER than the
        // plateau that are bistable -- the same build runs a whole bench at 0.227 ms per launch or at 0.267, and nothing in a short visit
        // tells which it will be.  So: drop what is more than 6 % above the fastest (the cliff), and of the rest hold the MEDIAN one --
        // neither on the cliff nor on the edge.  A placement that later reads > 4 % above its survey figure for CW_PLACE_BAD_WINDOWS
        // windows is struck off (a.place_struck) and the survey repeated.
        int best = a.place, n_cand = 0;
        float best_ms = 0.f, med[CW_PLACES];
        char log[256] = "";
        size_t len = 0;
        for (int k = 0; k < CW_PLACES; k++) {
            float *m = a.survey_ms[k];
            med[k] = 0.f;
            if (m[0] <= 0 || m[1] <= 0 || m[2] <= 0) continue;                   // (a lost sample: the placement does not compete)
            med[k] = std::max(std::min(m[0], m[1]), std::min(std::max(m[0], m[1]), m[2]));
            if (len < sizeof(log) - 16) len += (size_t)snprintf(log + len, sizeof(log) - len, " %d:%.4f%s", k, med[k], ((a.place_struck >> k) & 1) ? "x" : "");
        }
        const int pick = cwh_choose_place(med, a.place_struck, &n_cand);
        if (pick >= 0) { best = pick; best_ms = med[pick]; }
        if (verbose) fprintf(stderr, "[craftingworld] placement survey %u (window %u), median ms/step by placement:%s -> %d (median of the %d within 6 %% of the fastest)\n",
                             a.surveys, w, log, best, n_cand);
        a.place = best;
        a.place_ms = best_ms;
        a.place_bad = 0;
        a.surveying = false;
        for (int p = 0; p < 16; p++) a.stat[p] = 0;                              // the pace figures belonged to the old placement
    }
    if (a.place_on && !a.surveying && (a.surveys == 0 || a.place_bad >= CW_PLACE_BAD_WINDOWS)) {
        if (a.surveys) {
            if (verbose) fprintf(stderr, "[craftingworld] placement %d has read > 4 %% above its %.4f ms/step for %d windows: struck off, new survey\n",
                                 a.place, a.place_ms, a.place_bad);
            a.place_struck |= 1u << a.place;
            if ((a.place_struck & 0xFFu) == 0xFFu) a.place_struck = 0;              // (all struck: the regime has moved as a whole; start over)
        }
        a.surveying = true;
        a.survey_w0 = w;
        a.survey_seen = 0;
        a.place_bad = 0;
        a.surveys++;
        for (int k = 0; k < CW_PLACES; k++) a.survey_ms[k][0] = a.survey_ms[k][1] = a.survey_ms[k][2] = 0.f;
    }
    if (a.surveying && w < a.survey_w0 + CW_SURVEY_ROUNDS * CW_PLACES) {         // a survey window: placement by turns, pace frozen
        const unsigned j = w - a.survey_w0;
        a.place_of_window[w % CW_ADAPT_RING] = (signed char)(j % CW_PLACES);
        a.round_of_window[w % CW_ADAPT_RING] = (signed char)(j / CW_PLACES + 1);
        a.pace_of_window[w % CW_ADAPT_RING] = (signed char)a.cur;
        return;
    }
    a.place_of_window[w % CW_ADAPT_RING] = (signed char)a.place;                            // (survey windows still being read: the old placement meanwhile)
    a.round_of_window[w % CW_ADAPT_RING] = 0;
    if (moved && !a.surveying && a.pace_on) {        // move to a neighbour that is measurably better (figures older than ~3 cycles do not count)
        const int c = a.cur;
        auto fresh = [&](int p) { return p >= 0 && p <= CW_ADAPT_MAX && a.stat[p] > 0 && a.next_window - a.stat_window[p] < 80; };
        if (fresh(c)) {
            int best = c;
            if (fresh(c + 1) && a.stat[c + 1] < a.stat[best] * 0.993f) best = c + 1;
            if (fresh(c - 1) && a.stat[c - 1] < a.stat[best] * (best == c ? 0.993f : 1.0f)) best = c - 1;
            if (best != c) {
                if (verbose)
                    fprintf(stderr, "[craftingworld] sleeps beside resets (online, window %u): +%d %.4f ms/step | +%d %.4f | +%d %.4f -> +%d\n", w, c,
                            a.stat[c], c + 1, fresh(c + 1) ? a.stat[c + 1] : 0.0, c - 1, fresh(c - 1) ? a.stat[c - 1] : 0.0, best);
                a.cur = best;
            }
        }
    }
    // the cycle: 0-19 cur | 20 (settle), 21 cur + 1 | 22 (settle), 23 cur - 1; window 0 of the cycle settles too
    const unsigned pos = w % 24;
    int p = a.cur;
    const bool settle = (pos == 0 || pos == 20 || pos == 22) || a.surveying;
    if (!a.surveying && a.pace_on) {
        if (pos == 20 || pos == 21) p = a.cur + 1 > CW_ADAPT_MAX ? a.cur : a.cur + 1;
        else if (pos >= 22) p = a.cur > 0 ? a.cur - 1 : a.cur;
    }
    a.pace_of_window[w % CW_ADAPT_RING] = (signed char)(settle ? -1 - p : p);
}

// XCD-aware frame shares for the full-frame render kernel.  On MI355X the workgroups of every other XCD write ~15 % slower
// than their neighbours' (workgroups go round-robin over the 8 XCDs, so it shows as even vs odd workgroup index), and a
// launch lasts as long as its slowest wave.  Measure it instead of assuming it: a few equal-share launches with the waves'
// busy time summed per index parity, then the slow class paints q_all frames per wave and the fast class the rest.  Only
// performance depends on the outcome; which frames are painted does not (cw_kernels.hip: render_jobs).
static int calibrate_render_shares(cw_engine *e)
{
    CwTuning &tn = e->tune;
    tn.render_q_all = 0;
    tn.render_fast_parity = -1;
#ifdef CW_EXPERIMENT
    const char *off = getenv("CW_TUNE_RENDER_SHARES");
#else
    const char *off = nullptr;
#endif
    if (e->obs_mode != CW_OBS_PIXELS_FULL || e->host_actions || (off && atoi(off) == 0)) return CW_OK;   // (host-mapped frames: PCIe-bound anyway)
    // the paced linear sweep does not profit (m+1: 0.2349 / 0.2349 / 0.2356 ms with shares, 0.2355 / 0.2353 / 0.2349 with equal ones,
    // alternating on one box, profiles/history/r02_pace.txt): the shares are for the frame-per-wave kernel; CW_TUNE_RENDER_SHARES=1 forces them
    // (the one-launch step runs the sweep on equal shares whatever is calibrated here: nothing to measure for it)
    if (cwk_render_is_linear(&e->P, &tn) && (!(off && atoi(off) != 0) || cwk_step_renders_fused(&e->P, &tn, e->auto_reset))) return CW_OK;
#ifdef CW_EXPERIMENT
    const char *forced_shares = getenv("CW_TUNE_RENDER_QALL");              // "q_all,parity"
#else
    const char *forced_shares = nullptr;
#endif
    if (const char *q = forced_shares) {
        int qa = 0, par = -1;
        if (sscanf(q, "%d,%d", &qa, &par) == 2 && qa > 0 && (par == 0 || par == 1)) { tn.render_q_all = qa; tn.render_fast_parity = par; }
        return CW_OK;
    }
    unsigned long long *stats = nullptr;
    int rc = dev_alloc(e, &stats, 2);
    if (rc != CW_OK) return rc;
    int blocks = 0, wpb = 0;
    e->P.render_stats = stats;
    // busy[c] = summed busy time of the waves of workgroup-index parity c over 3 launches, after 3 warm-up launches, all queued
    // back to back (see calibrate_render_pace: no host round trip between launches)
    auto measure = [&](int q_all, int parity, double busy[2]) -> int {
        unsigned long long h[2] = {0, 0};
        bool ok = true;
        for (int rep = 0; rep < 6 && ok; rep++) {
            if (rep == 3) ok = hipMemsetAsync(stats, 0, sizeof(h), nullptr) == hipSuccess;
            ok = ok && cwk_launch_render_calib(&e->P, &tn, nullptr, q_all, parity, &blocks, &wpb) == hipSuccess;
        }
        if (!ok || hipDeviceSynchronize() != hipSuccess || hipMemcpy(h, stats, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess)
            return fail(CW_ERR_HIP, "cw_create: render calibration failed");
        busy[0] = (double)h[0];
        busy[1] = (double)h[1];
        return CW_OK;
    };
    const int jobs = cwk_render_jobs(&e->P, &tn);              // frames, or (frame, row group) pairs of the linear sweep
    double busy[2], busy0[2];
    rc = measure(0, -1, busy0);                                  // equal shares
    int q_all = 0, fast = -1;
    long long n_half = 0;
    if (rc == CW_OK && blocks >= 16 && !(blocks & 1) && busy0[0] > 0 && busy0[1] > 0) {
        n_half = (long long)blocks * wpb / 2;
        fast = busy0[1] < busy0[0] ? 1 : 0;
        double rho = busy0[1 - fast] / busy0[fast];              // per-frame cost of the slow class relative to the fast one
        if (rho >= 1.03) {
            // the classes share the memory system, so shifting frames changes both costs: refine on what is measured
            for (int it = 0; it < 3 && rc == CW_OK; it++) {
                if (rho > 1.6) rho = 1.6;
                q_all = (int)((double)jobs / ((double)n_half * (1.0 + rho)));
                if (q_all < 1) { q_all = 0; break; }
                rc = measure(q_all, fast, busy);
                if (rc != CW_OK || busy[0] <= 0 || busy[1] <= 0) break;
                const double q_fast = ((double)jobs - (double)q_all * (double)n_half) / (double)n_half;
                const double per_frame_slow = busy[1 - fast] / (double)q_all, per_frame_fast = busy[fast] / q_fast;
                rho = per_frame_slow / per_frame_fast;
                if (rho < 1.0) rho = 1.0;
            }
            if (q_all >= 1) q_all = (int)((double)jobs / ((double)n_half * (1.0 + (rho > 1.6 ? 1.6 : rho))));
        }
    }
    e->P.render_stats = nullptr;
    if (rc != CW_OK) return rc;
    if (q_all < 1) return CW_OK;                                 // small batch, or balanced already: equal shares
    tn.render_q_all = q_all;
    tn.render_fast_parity = fast;
    const long long n_waves = 2 * n_half;
    busy[0] = busy0[0];
    busy[1] = busy0[1];
    if (getenv("CW_TUNE_VERBOSE"))
        fprintf(stderr, "[craftingworld] render shares: waves of even/odd workgroups busy %.1f / %.1f us on equal shares -> %d jobs per slow wave, "
                "%s workgroups take the rest\n", busy[0] / 3.0 / (double)(n_waves / 2) / 100.0, busy[1] / 3.0 / (double)(n_waves / 2) / 100.0,
                q_all, fast ? "odd" : "even");
    return CW_OK;
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
    P.task_mask = (1u << cfg->n_task_list) - 1u;
    P.pool_k = e->K;
    P.div_magic = (uint32_t)((1ull << 32) / (uint64_t)e->S) + 1u;
    P.raster = cfg->raster;
    P.frame_bytes = cfg->raster == CW_RASTER_ALT ? 27u * (uint32_t)e->S * (uint32_t)(e->S + 1) : 48u * (uint32_t)e->ncell;
    P.grp_rows = e->S <= 64 ? 64 / e->S : 0;
    P.grp_per_frame = P.grp_rows ? (e->S + P.grp_rows - 1) / P.grp_rows : 0;
    {   // Tuning.  The defaults are the measured best; a product build reads six environment variables -- CW_TUNE_VERBOSE,
        // CW_TUNE_RENDER_ADAPT, CW_TUNE_RENDER_PLACE, CW_TUNE_RENDER_PACE, CW_TUNE_RENDER_PACE_BESIDE, CW_TUNE_RENDER_CHUNK_ROUNDS -- and a
        // build with -DCW_EXPERIMENT (libcraftingworld_exp.so: A/B runs and the tests that hold the older launch arrangements to the same
        // results) the launch-shape knobs below as well.
        auto geti = [](const char *k, int d) { const char *v = getenv(k); return v ? atoi(v) : d; };
        CwTuning &tn = e->tune;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) tn.n_cu = prop.multiProcessorCount;
        tn.list_blocks = tn.n_cu;
        P.tune_reset_prio = 2;                               // the render waves raise their priority, the reset kernel beside them does not
        tn.render_chunk_rounds = geti("CW_TUNE_RENDER_CHUNK_ROUNDS", tn.render_chunk_rounds);
        tn.render_place = geti("CW_TUNE_RENDER_PLACE", tn.render_place) & 7;
        tn.piece_sweep = geti("CW_TUNE_PIECE_SWEEP", tn.piece_sweep);
#ifdef CW_EXPERIMENT
        P.tune_reset_prio = geti("CW_TUNE_RESET_PRIO", 2);
        tn.render_blocks_per_cu = geti("CW_TUNE_RENDER_BLOCKS_PER_CU", tn.render_blocks_per_cu);
        tn.render_blocks_abs = geti("CW_TUNE_RENDER_BLOCKS", tn.render_blocks_abs);
        const int rt = geti("CW_TUNE_RENDER_THREADS", tn.render_threads);
        if (rt == 64 || rt == 128 || rt == 256) tn.render_threads = rt;
        tn.list_blocks = geti("CW_TUNE_LIST_BLOCKS", tn.list_blocks);
        tn.overlap = geti("CW_TUNE_OVERLAP", tn.overlap);
        tn.fused_step = geti("CW_TUNE_FUSED_STEP", tn.fused_step);
        tn.profile_side = geti("CW_PROFILE_SIDE_STREAM", tn.profile_side);
        tn.render_linear = geti("CW_TUNE_RENDER_LINEAR", tn.render_linear);
        tn.render_pace_fine = geti("CW_TUNE_RENDER_FINE", tn.render_pace_fine);
        tn.fused_render = geti("CW_TUNE_FUSED_RENDER", tn.fused_render);
        tn.reset_blocks_per_cu = geti("CW_TUNE_RESET_BLOCKS_PER_CU", tn.reset_blocks_per_cu);
        tn.fused_reset_blocks_per_cu = geti("CW_TUNE_FUSED_RESET_BLOCKS_PER_CU", tn.fused_reset_blocks_per_cu);
#endif
        if (tn.reset_blocks_per_cu < 1) tn.reset_blocks_per_cu = 1;
        if (tn.reset_blocks_per_cu > 8) tn.reset_blocks_per_cu = 8;
        if (tn.fused_reset_blocks_per_cu < 1) tn.fused_reset_blocks_per_cu = 1;
        if (tn.fused_reset_blocks_per_cu > 8) tn.fused_reset_blocks_per_cu = 8;
        if (tn.render_blocks_per_cu < 1) tn.render_blocks_per_cu = 1;
        if (tn.list_blocks < 1) tn.list_blocks = 1;
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
    ALLOC(done_list, N);
    ALLOC(done_count, 2);
    ALLOC(counters, 4);
    if (cfg->obs_mode != CW_OBS_STATE) {
        ALLOC_OUT(obs, N * P.frame_bytes);
        ALLOC_OUT(desired_img, N * P.frame_bytes);
        ALLOC_OUT(init_img, N * P.frame_bytes);
        if (cfg->keep_terminal_obs && cfg->auto_reset) ALLOC_OUT(terminal_img, N * P.frame_bytes);
    }
    if (cfg->host_outputs && rc == CW_OK) rc = host_alloc(e, &e->host_actions, N);
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
    if (rc == CW_OK && (hipStreamCreateWithFlags(&e->side, hipStreamNonBlocking) != hipSuccess ||
                        hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming) != hipSuccess ||
                        hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming) != hipSuccess))
        rc = fail(CW_ERR_HIP, "cw_create: side stream / event creation failed");
    if (rc != CW_OK) {
        for (void *p : e->allocs) (void)hipFree(p);
        for (void *p : e->host_allocs) (void)hipHostFree(p);
        if (e->side) (void)hipStreamDestroy(e->side);
        if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
        if (e->ev_join) (void)hipEventDestroy(e->ev_join);
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
    const int want_piece_sweep = e->tune.piece_sweep;
    e->tune.piece_sweep = 0;                                       // (the frame-per-wave painter first: its pace and shares also serve the frames of resets)
    if (rc == CW_OK) rc = calibrate_render_pace(e, false);
    if (rc == CW_OK) rc = calibrate_render_shares(e);
    if (rc == CW_OK && e->tune.render_fast_parity >= 0) rc = calibrate_render_pace(e, true);
    e->tune.piece_sweep = want_piece_sweep;
    if (rc == CW_OK) rc = calibrate_piece_sweep(e);
    if (rc == CW_OK) e->tune.render_pace |= (e->tune.render_pace_fine & 0xFF) << 16;
    if (rc == CW_OK && e->obs_mode == CW_OBS_PIXELS_FULL && e->auto_reset && !e->host_actions &&
        (cwk_render_is_linear(&e->P, &e->tune) || cwk_render_is_piece_sweep(&e->P, &e->tune)) && !(getenv("CW_TUNE_RENDER_ADAPT") && atoi(getenv("CW_TUNE_RENDER_ADAPT")) == 0) && (long long)e->n * e->P.frame_bytes >= (64ll << 20)) {
        cw_engine::Adapt &a = e->adapt;
        a.cur = (e->tune.render_pace >> 12) & 15;
        if (a.cur > CW_ADAPT_MAX) a.cur = CW_ADAPT_MAX;
        a.pace_on = !getenv("CW_TUNE_RENDER_PACE_BESIDE");
        a.place = e->tune.render_place;
        a.place_on = !getenv("CW_TUNE_RENDER_PLACE") && cwk_step_renders_fused(&e->P, &e->tune, e->auto_reset) && cwk_render_is_linear(&e->P, &e->tune) &&
                     !cwk_render_is_piece_sweep(&e->P, &e->tune);          // (the placements are render_groups': nothing to survey for the sweep of pieces)
        // (3) only the Ray raster's unpaced piece sweep (AltObs runs paced already), and only if nobody forced a pace
        a.guard_on = cwk_render_is_piece_sweep(&e->P, &e->tune) && e->P.raster != CW_RASTER_ALT && e->tune.piece_pace == 0 && !getenv("CW_TUNE_PIECE_PACE") &&
                     !(getenv("CW_TUNE_REGIME_GUARD") && atoi(getenv("CW_TUNE_REGIME_GUARD")) == 0);
#ifdef CW_EXPERIMENT
        a.guard_test = a.guard_on && getenv("CW_TUNE_REGIME_GUARD") && atoi(getenv("CW_TUNE_REGIME_GUARD")) == 2;
#endif
        if (a.guard_on && !a.guard_test) a.guard[2] = 3.f;                    // (the opening survey)
        if (a.pace_on || a.place_on || a.guard_on) {
            for (hipEvent_t &ev : a.ev)
                if (rc == CW_OK && hipEventCreate(&ev) != hipSuccess) rc = fail(CW_ERR_HIP, "cw_create: tuner set-up failed");
            a.on = rc == CW_OK;
        }
    }
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
    (void)hipDeviceSynchronize();
    prof_free(e);
    if (e->res_stream) (void)hipStreamDestroy(e->res_stream);
    if (e->side) (void)hipStreamDestroy(e->side);
    if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
    if (e->ev_join) (void)hipEventDestroy(e->ev_join);
    for (hipEvent_t ev : e->adapt.ev) if (ev) (void)hipEventDestroy(ev);
    for (void *p : e->allocs) (void)hipFree(p);
    for (void *p : e->host_allocs) (void)hipHostFree(p);
    delete e;
    return CW_OK;
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
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(e->P.mt, keys, N * CW_MT_N * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(e->P.mt_idx, pos, N * sizeof(int32_t), hipMemcpyHostToDevice));
    HIP_TRY(cwk_launch_seed(&e->P, nullptr, nullptr));
    HIP_TRY(hipDeviceSynchronize());
    return CW_OK;
}

int cw_seed_int(cw_engine *e, const uint32_t *seeds)
{
    if (!e || !seeds) return fail(CW_ERR_INVALID, "cw_seed_int: null argument");
    DeviceGuard guard(e->device);
    if (!guard.ok) return fail(CW_ERR_HIP, "hipSetDevice(%d) failed", e->device);
    PARK(e);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(e->seed_scratch, seeds, (size_t)e->n * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIP_TRY(cwk_launch_seed(&e->P, e->seed_scratch, nullptr));
    HIP_TRY(hipDeviceSynchronize());
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
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(words.data(), e->P.mt, words.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(pos, e->P.mt_idx, N * sizeof(int32_t), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < N; i++) cwh_mt_to_numpy(&words[i * CW_MT_N], pos[i], keys + i * CW_MT_N);
    return CW_OK;
}

int cw_generate_fixed_states(cw_engine *e, cw_stream_t stream)
{
    if (!e) return fail(CW_ERR_INVALID, "cw_generate_fixed_states: null engine");
    if (e->K == 0) return CW_OK;
    DeviceGuard guard(e->device);
    if (!guard.ok) return fail(CW_ERR_HIP, "hipSetDevice(%d) failed", e->device);
    PARK(e);
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
    HIP_TRY(cwk_launch_reset_all(&e->P, &e->tune, e->obs_mode, (hipStream_t)stream));
    e->has_reset = true;
    e->last_stream = (hipStream_t)stream; e->last_stream_set = true;
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
    e->last_stream = (hipStream_t)stream; e->last_stream_set = true;
    hipEvent_t *ev = (e->prof_n < e->prof_cap) ? &e->prof_ev[(size_t)e->prof_n * 6] : nullptr;
    if (e->adapt.on) {                               // full-frame mode: the sweep's extra sleeps beside resets follow what the steps measure (adapt_tick)
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing((hipStream_t)stream, &cap) != hipSuccess || cap == hipStreamCaptureStatusNone) {
            if (e->adapt.seq % CW_ADAPT_W == 0) adapt_tick(e, (hipStream_t)stream);
            const int pw = e->adapt.pace_of_window[(e->adapt.seq / CW_ADAPT_W) % CW_ADAPT_RING];
            e->tune.render_pace = (e->tune.render_pace & 0xFF01FF) | ((pw < 0 ? -1 - pw : pw) << 12);
            if (e->adapt.place_on) e->tune.render_place = e->adapt.place_of_window[(e->adapt.seq / CW_ADAPT_W) % CW_ADAPT_RING];
            if (ev) e->adapt.tainted[(e->adapt.seq / CW_ADAPT_W) % CW_ADAPT_RING] = true;
            e->adapt.seq++;
        } else {
            e->tune.render_pace = (e->tune.render_pace & 0xFF01FF) | (e->adapt.cur << 12);     // a captured graph keeps the values it was captured with
            if (e->adapt.place_on) e->tune.render_place = e->adapt.place;
        }
    }
    HIP_TRY(cwk_launch_step(&e->P, &e->tune, actions, action_dtype, e->obs_mode, e->auto_reset, (hipStream_t)stream, e->side,
                            e->ev_fork, e->ev_join, ev));
    if (ev) e->prof_n++;
    return CW_OK;
}

// One step of the single-env loop WITHOUT a kernel launch: ring the resident kernel's doorbell, spin on its answer (cw_kernels.hip:
// cw_resident_kernel).  The kernel is (re)launched on demand -- the first call, after it idled out (2 ms without a request), after its
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
            // (the new instance reads the env's records: whatever cw_reset / cw_step enqueued last on the caller's stream comes first)
            if (e->last_stream_set) { HIP_TRY(hipStreamSynchronize(e->last_stream)); e->last_stream_set = false; }
            __atomic_store_n(&R->exited, 0u, __ATOMIC_RELEASE);
            __atomic_store_n(&R->ack, (seq - 1u) & 0xFFFFFFu, __ATOMIC_RELEASE);
            HIP_TRY(cwk_launch_resident(&e->P, R, (seq - 1u) & 0xFFFFFFu, e->obs_mode == CW_OBS_PIXELS_DIRTY ? 1 : 0,
                                        200000ull /* 2 ms idle */, 20000000ull /* 200 ms slice */, e->res_stream));
            e->res_running = true;
            continue;
        }
        __builtin_ia32_pause();
        if ((spins & 1023u) == 1023u) {
            struct timespec t1;
            clock_gettime(CLOCK_MONOTONIC, &t1);
            if ((t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec) > 5.0) {
                (void)resident_park(e);
                return fail(CW_ERR_HIP, "cw_step_resident: no answer from the resident kernel within 5 s (parked)");
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
    e->last_stream = (hipStream_t)stream; e->last_stream_set = true;
    HIP_TRY(cwk_launch_rollout(&e->P, actions, n_steps, rewards, dones, (hipStream_t)stream));
    return CW_OK;
}

int cw_render(cw_engine *e, uint8_t *out_frames, cw_stream_t stream)
{
    if (!e || !out_frames) return fail(CW_ERR_INVALID, "cw_render: null argument");
    if (!e->has_reset) return fail(CW_ERR_STATE, "cw_render called before cw_reset");
    DeviceGuard guard(e->device);
    if (!guard.ok) return fail(CW_ERR_HIP, "hipSetDevice(%d) failed", e->device);
    PARK(e);
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
    HIP_TRY(cwk_launch_render_onehot(&e->P, onehot, n_states, out_frames, (hipStream_t)stream));
    return CW_OK;
}

int cw_export_grid(cw_engine *e, uint8_t *out, cw_stream_t stream)
{
    if (!e || !out) return fail(CW_ERR_INVALID, "cw_export_grid: null argument");
    DeviceGuard guard(e->device);
    if (!guard.ok) return fail(CW_ERR_HIP, "hipSetDevice(%d) failed", e->device);
    PARK(e);
    HIP_TRY(cwk_launch_export(&e->P, &e->tune, out, 0, 0, (hipStream_t)stream));
    return CW_OK;
}

int cw_export_onehot(cw_engine *e, uint8_t *out, cw_stream_t stream)
{
    if (!e || !out) return fail(CW_ERR_INVALID, "cw_export_onehot: null argument");
    DeviceGuard guard(e->device);
    if (!guard.ok) return fail(CW_ERR_HIP, "hipSetDevice(%d) failed", e->device);
    PARK(e);
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
        const bool side_recorded = e->tune.profile_side || !(e->obs_mode == CW_OBS_PIXELS_FULL && e->auto_reset && e->tune.overlap);
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
    if (e->obs_mode == CW_OBS_PIXELS_DIRTY) return e->auto_reset && e->tune.fused_step ? "cw_step_fused_kernel" : "cw_step_kernel";
    const bool one_launch = cwk_step_renders_fused(&e->P, &e->tune, e->auto_reset);
    if (cwk_render_is_piece_sweep(&e->P, &e->tune)) return one_launch ? "cw_render_pieces_step_kernel" : "cw_render_pieces_kernel";
    if (!cwk_render_is_linear(&e->P, &e->tune)) return one_launch ? "cw_render_frames_step_kernel" : "cw_render_frames_kernel";
    return one_launch ? "cw_render_step_kernel" : "cw_render_kernel";
}

int cw_tuner(const cw_engine *e, cw_tuner_state *out)
{
    if (!e || !out) return fail(CW_ERR_INVALID, "cw_tuner: null argument");
    const cw_engine::Adapt &a = e->adapt;
    out->place = a.place_on ? a.place : e->tune.render_place;
    out->surveys = (int32_t)a.surveys;
    out->struck_mask = (int32_t)a.place_struck;
    out->sleeps_beside = a.on && a.pace_on ? a.cur : (e->tune.render_pace >> 12) & 15;
    out->place_tuned = a.on && a.place_on ? 1 : 0;
    out->sleeps_tuned = a.on && a.pace_on ? 1 : 0;
    out->painter = cwk_render_is_piece_sweep(&e->P, &e->tune) ? 2 : cwk_render_is_linear(&e->P, &e->tune) ? 1 : 0;
    out->piece_pace = e->tune.piece_pace;
    out->guard_state = a.on && a.guard_on ? (int32_t)a.guard[2] : -1;
    out->guard_trials = a.guard_trials;
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
    return CW_OK;
}

// ------------------------------------------------------------------------------ state get/set
static void slots_to_grid(const uint16_t *pos, uint32_t codes, int ncell, uint8_t *grid)
{
    memset(grid, 0, (size_t)ncell);
    for (int k = 0; k < 8; k++)
        if (pos[k] < (uint32_t)ncell) grid[pos[k]] = (uint8_t)((codes >> (4 * k)) & 15u);
}

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
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(hdr.data(), e->P.hdr, N * 16, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(pos.data(), e->P.pos, N * 16, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(ipos.data(), e->P.init_pos, N * 16, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(gpos.data(), e->P.goal_pos, N * 16, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(goal_codes.data(), e->P.goal_codes, N * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(iagent.data(), e->P.init_agent, N * 2, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(gagent.data(), e->P.goal_agent, N * 2, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(epno.data(), e->P.ep_no, N * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < N; i++) {
        const uint32_t *h = &hdr[i * 4];
        if (v->grid) slots_to_grid(&pos[i * 8], h[3], nc, v->grid + i * nc);
        if (v->init_grid) slots_to_grid(&ipos[i * 8], CW_CODES_INITIAL, nc, v->init_grid + i * nc);
        if (v->goal_grid) slots_to_grid(&gpos[i * 8], goal_codes[i], nc, v->goal_grid + i * nc);
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
    HIP_TRY(hipDeviceSynchronize());
    if (restore_episode) {                           // the episode records: goal state (imagine_obs' result) and the agent's start cell
        gpos.resize(N * 8); gagent.resize(N); iagent.resize(N); gcodes.resize(N);
        HIP_TRY(hipMemcpy(gpos.data(), e->P.goal_pos, N * 16, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(gcodes.data(), e->P.goal_codes, N * 4, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(gagent.data(), e->P.goal_agent, N * 2, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(iagent.data(), e->P.init_agent, N * 2, hipMemcpyDeviceToHost));
    }
    HIP_TRY(hipMemcpy(hdr.data(), e->P.hdr, N * 16, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(pos.data(), e->P.pos, N * 16, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(ipos.data(), e->P.init_pos, N * 16, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(epno.data(), e->P.ep_no, N * 4, hipMemcpyDeviceToHost));
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
    HIP_TRY(hipMemcpy(e->P.hdr, hdr.data(), N * 16, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(e->P.pos, pos.data(), N * 16, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(e->P.init_pos, ipos.data(), N * 16, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(e->P.ep_no, epno.data(), N * 4, hipMemcpyHostToDevice));
    if (restore_episode) {
        HIP_TRY(hipMemcpy(e->P.goal_pos, gpos.data(), N * 16, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(e->P.goal_codes, gcodes.data(), N * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(e->P.goal_agent, gagent.data(), N * 2, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(e->P.init_agent, iagent.data(), N * 2, hipMemcpyHostToDevice));
    }
    if (e->obs_mode != CW_OBS_STATE) {   // the persistent frames must follow the injected state
        if (restore_episode) HIP_TRY(cwk_launch_render_restore(&e->P, &e->tune, nullptr));
        else HIP_TRY(cwk_launch_render_ext(&e->P, &e->tune, e->P.obs, nullptr));
    }
    HIP_TRY(hipDeviceSynchronize());
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
    uint32_t task_mask, n_menus;
    uint64_t menus_hash, total_bytes;
};
static const char CW_CKPT_MAGIC[8] = {'C', 'W', 'C', 'K', 'P', 'T', 0, 1};

struct CkptSection { void *dev; size_t bytes; };
static std::vector<CkptSection> ckpt_sections(cw_engine *e)
{
    const CwParams &P = e->P;
    const size_t N = (size_t)e->n;
    return {{P.hdr, N * 16}, {P.pos, N * 16}, {P.init_pos, N * 16}, {P.goal_pos, N * 16}, {P.goal_codes, N * 4},
            {P.init_agent, N * 2}, {P.goal_agent, N * 2}, {P.ep_no, N * 4}, {P.mt, N * CW_MT_N * 4}, {P.mt_idx, N * 4},
            {P.pool, N * (size_t)e->K * 9 * 2}, {P.reward, N * 4}, {P.done, N}, {P.achieved_out, N * 2}, {P.desired_out, N * 2},
            {P.episode_length, N * 4}, {P.counters, 4 * 8}};
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
    h.version = 1;
    h.header_bytes = (uint32_t)sizeof(CwCkptHeader);
    h.n_envs = e->n; h.size = e->S; h.max_steps = e->P.max_steps; h.pool_k = e->K;
    h.task_mask = e->P.task_mask; h.n_menus = (uint32_t)e->menus.size();
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
    HIP_TRY(hipDeviceSynchronize());
    unsigned char *p = (unsigned char *)buf;
    memcpy(p, &h, sizeof(h));
    p += sizeof(h);
    for (const CkptSection &sec : ckpt_sections(e)) {
        if (sec.bytes) HIP_TRY(hipMemcpy(p, sec.dev, sec.bytes, hipMemcpyDefault));
        p += sec.bytes;
    }
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
    if (memcmp(h.magic, CW_CKPT_MAGIC, 8) != 0 || h.version != 1 || h.header_bytes != sizeof(CwCkptHeader))
        return fail(CW_ERR_INVALID, "cw_checkpoint_load: not a CraftingWorld checkpoint (or another version)");
    if (h.n_envs != mine.n_envs || h.size != mine.size || h.max_steps != mine.max_steps || h.pool_k != mine.pool_k ||
        h.task_mask != mine.task_mask)
        return fail(CW_ERR_INVALID, "cw_checkpoint_load: checkpoint of %d envs, size %d, max_steps %d, fixed_init_state %d, task mask %#x; "
                    "this engine: %d, %d, %d, %d, %#x", h.n_envs, h.size, h.max_steps, h.pool_k, h.task_mask, mine.n_envs, mine.size,
                    mine.max_steps, mine.pool_k, mine.task_mask);
    if (h.n_menus != mine.n_menus || h.menus_hash != mine.menus_hash)
        return fail(CW_ERR_INVALID, "cw_checkpoint_load: the checkpoint was written with different task menus (selected_tasks / "
                    "number_of_tasks / stacking / reward_style)");
    if (h.total_bytes != mine.total_bytes || length < h.total_bytes)
        return fail(CW_ERR_INVALID, "cw_checkpoint_load: truncated checkpoint (%zu of %llu bytes)", length, (unsigned long long)h.total_bytes);
    HIP_TRY(hipDeviceSynchronize());
    const unsigned char *p = (const unsigned char *)buf + sizeof(h);
    for (const CkptSection &sec : ckpt_sections(e)) {
        if (sec.bytes) HIP_TRY(hipMemcpy(sec.dev, p, sec.bytes, hipMemcpyDefault));
        p += sec.bytes;
    }
    e->has_reset = true;
    if (e->obs_mode != CW_OBS_STATE) HIP_TRY(cwk_launch_render_restore(&e->P, &e->tune, nullptr));   // frames follow the records
    HIP_TRY(hipDeviceSynchronize());
    return CW_OK;
}

}  // extern "C"
