// cw_host.cpp -- the engine's HIP-free host logic (cw_host.h): no HIP header, no engine struct.  Built into libcraftingworld.so, and alone under
// ASAN/UBSAN for the CPU test tier (make host_asan).
#include "cw_host.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>

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
    if (pos > CWH_MT_N) pos = CWH_MT_N;
    for (int k = 0; k < pos; k++)
        s[k] = mt_twist(s[k], s[(k + 1) % CWH_MT_N], s[(k + 397) % CWH_MT_N]);
    return pos % CWH_MT_N;
}

// inverse: from engine form (s, idx) recover a numpy key whose stream from position idx is
// identical.  Words < idx are un-twisted; key[0]'s low 31 bits are not part of the MT19937 state and
// cannot be un-twisted ...
void cwh_mt_to_numpy(const uint32_t *s, int idx, uint32_t *key)
{
    for (int j = idx; j < CWH_MT_N; j++) key[j] = s[j];
    for (int j = 0; j < idx; j++) key[j] = 0;
    for (int j = idx - 1; j >= 0; j--) {
        const uint32_t m = (j < CWH_MT_N - 397) ? key[j + 397] : s[j - (CWH_MT_N - 397)];
        uint32_t t = s[j] ^ m;
        const uint32_t odd = t >> 31;
        if (odd) t ^= 0x9908b0dfu;
        const uint32_t y = (t << 1) | odd;     // (G[j] & UPPER) | (G[j+1] & LOWER)
        key[j] |= y & 0x80000000u;
        if (j + 1 < idx) key[j + 1] |= y & 0x7fffffffu;
    }
    if (idx > 0) {     // ... but they are what the generation's word 623 was made with: key[623] = key[396] ^ T(previous[623].hi, key[0].lo)
        uint32_t t = key[CWH_MT_N - 1] ^ key[396];
        const uint32_t odd = t >> 31;
        if (odd) t ^= 0x9908b0dfu;
        key[0] |= ((t << 1) | odd) & 0x7fffffffu;      // (exact for every key a twist produced; a rewind to position 0 reads this word again)
    }
}

// One generation back: key = all 624 words of a generation as numpy holds them after its twist -> the generation before (whose twist
// produced it).  The twist is a bijection on the 19 937 state bits (word 0 counts with its top bit only): word k >= 227 of the new
// generation is new[k-227] ^ T(old[k].hi, old[k+1].lo), word 623 new[396] ^ T(old[623].hi, new[0].lo), word k < 227 old[k+397] ^ T(...);
// T(y) = (y >> 1) ^ (y odd ? 0x9908b0df : 0) is undone through its top bit.  The low 31 bits of key[0] (not part of the state; the export
// above restores them from words 623 and 396) are REPAIRED on the way in -- the step for word 227 needs them -- and restored in the result.
void cwh_mt_untwist(uint32_t *key)
{
    uint32_t prev[CWH_MT_N];
    memset(prev, 0, sizeof(prev));
    for (int k = CWH_MT_N - 1; k >= 0; k--) {
        uint32_t t = key[k] ^ (k == CWH_MT_N - 1 ? key[396] : k >= CWH_MT_N - 397 ? key[k - (CWH_MT_N - 397)] : prev[k + 397]);
        const uint32_t odd = t >> 31;
        if (odd) t ^= 0x9908b0dfu;
        const uint32_t y = (t << 1) | odd;
        prev[k] |= y & 0x80000000u;
        if (k == CWH_MT_N - 1) key[0] = (key[0] & 0x80000000u) | (y & 0x7fffffffu);
        else prev[k + 1] |= y & 0x7fffffffu;
    }
    uint32_t t = prev[CWH_MT_N - 1] ^ prev[396];          // prev[0]'s own low bits, the same way (cwh_mt_to_numpy): a rewind may stop at position 0
    const uint32_t odd = t >> 31;
    if (odd) t ^= 0x9908b0dfu;
    prev[0] |= ((t << 1) | odd) & 0x7fffffffu;
    memcpy(key, prev, sizeof(prev));
}
// numpy state (key, pos) -> the state n raw draws earlier (a look-ahead record's draws, cw_get_mt); pos stays in 0..623 like the export's
void cwh_mt_rewind(uint32_t *key, int32_t *pos, uint32_t n)
{
    while (n > 0) {
        if ((uint32_t)*pos >= n) { *pos -= (int32_t)n; n = 0; }
        else { n -= (uint32_t)*pos; cwh_mt_untwist(key); *pos = CWH_MT_N; }
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

void cwh_mt_init_genrand(uint32_t *s, uint32_t seed)   // numpy RandomState(int): init_genrand
{
    s[0] = seed;
    for (int i = 1; i < CWH_MT_N; i++) s[i] = 1812433253u * (s[i - 1] ^ (s[i - 1] >> 30)) + (uint32_t)i;
}

}  // extern "C"


extern "C" {

// ------------------------------------------------------------------------------ dense view of a slot record
void cwh_slots_to_grid(const uint16_t *pos, uint32_t codes, int ncell, uint8_t *grid)
{
    memset(grid, 0, (size_t)ncell);
    for (int k = 0; k < 8; k++)
        if (pos[k] < (uint32_t)ncell) grid[pos[k]] = (uint8_t)((codes >> (4 * k)) & 15u);
}

// ------------------------------------------------------------------------------ checkpoint blob sections (cw_engine.cpp: ckpt_sections pairs them with device pointers)
// hdr, pos, init_pos, goal_pos (16 B / env), goal_codes (4), init_agent, goal_agent (2), ep_no (4), mt (624 x 4), mt_idx (4), pool (k x 9 x 2),
// reward (4), done (1), achieved_out, desired_out (2), episode_length, episode_return (4), counters (5 x 8: the four public ones + the sweep's
// private word), then the look-ahead records verbatim -- la_depth of them per env (0: the engine keeps none): nx_init_pos, nx_goal_pos, nx_misc (16 each per
// record), nx_ctl (4: the ring's head and the QUEUED bit)
int cwh_ckpt_section_bytes(int64_t n, int32_t k, int32_t la_depth, size_t *sizes, uint64_t *total)
{
    const size_t N = n > 0 ? (size_t)n : 0, K = k > 0 ? (size_t)k : 0, D = la_depth > 0 ? (size_t)la_depth : 0, la = D ? 1 : 0;
    const size_t sz[CWH_CKPT_SECTIONS] = {N * 16, N * 16, N * 16, N * 16, N * 4, N * 2, N * 2, N * 4, N * CWH_MT_N * 4, N * 4, N * K * 9 * 2,
                                          N * 4, N, N * 2, N * 2, N * 4, N * 4, 5 * 8,
                                          D * N * 16, D * N * 16, D * N * 16, la * N * 4};
    uint64_t sum = 0;
    for (int i = 0; i < CWH_CKPT_SECTIONS; i++) { if (sizes) sizes[i] = sz[i]; sum += sz[i]; }
    if (total) *total = sum;
    return CWH_CKPT_SECTIONS;
}

// ------------------------------------------------------------------------------ the sweep clock's periods and schedule
static double period_ns(int32_t sweep_waves, double tb_per_s) { return (double)sweep_waves * 4096.0 / (tb_per_s * 1e12) * 1e9; }
void cwh_sweep_periods(double rate, int32_t sweep_waves, double head_notch, double busy_notch, int32_t *period16, int32_t *period16_head, int32_t *period16_busy)
{
    const double head = rate - head_notch > CWH_GUARD_RATE_FLOOR ? rate - head_notch : CWH_GUARD_RATE_FLOOR;
    const double busy = rate - busy_notch > CWH_GUARD_RATE_FLOOR ? rate - busy_notch : CWH_GUARD_RATE_FLOOR;
    *period16 = rate > 0 ? (int32_t)(period_ns(sweep_waves, rate) * 1.6 + 0.5) : 0;
    *period16_head = rate > 0 ? (int32_t)(period_ns(sweep_waves, head) * 1.6 + 0.5) : 0;
    *period16_busy = rate > 0 ? (int32_t)(period_ns(sweep_waves, busy) * 1.6 + 0.5) : 0;
}
double cwh_guard_scheduled_ms(double sweep_jobs, int32_t period16, int32_t period16_busy, double beside_ms)
{   // (as after a step on which envs finished: a quiet step is 4 us early)
    return (sweep_jobs * (period16 / 1.6) + CWH_HEAD_JOBS * ((period16_busy - period16) / 1.6)) * 1e-6 + beside_ms;
}

// ------------------------------------------------------------------------------ the guard's decisions
// Every CW_GUARD_EVERY-th step's sweep is timed (cw_engine.cpp) and held against its schedule -- jobs x period + the busy head + what a launch costs
// beside its jobs (measured at cw_create).  A sweep in the memory system's saturated regime misses that by 10-16 % launch after launch; at the edge
// (7.7 TB/s) one launch in ten is 7-12 % late and the rest on time.  Three samples in a row more than 6 % late: the rate goes down by a notch (a
// SLOWDOWN: the only move that is not a trial).
// TRIALS (round 5).  cw_create's choice is a measurement of one moment: an engine created while the card was in a worse state settles a notch or two
// under what the card takes an hour later, and round 4's guard only ever went down.  Every move UP is a trial: after enough samples on time
// (recover_need below the best rate known, probe_need at it: a PROBE, never beyond the ceiling, the write path's edge) the guard tries ONE notch more
// and keeps it only if it PAYS -- the mean of CWH_GUARD_PROBE_SAMPLES sweeps at the new rate must be under the mean at the old one; "on time" is
// not enough (a clock a little too fast is on time and slower) and not needed either: with something else between the sweeps (another engine's step
// kernel) every sweep is a constant late, a rate judged by its schedule alone comes to rest a notch or two under the one with the shortest sweeps, and
// so a rate a trial has accepted is from then on measured against what it delivered then (ref_ms).  A trial that does not pay is undone and the next
// one of its kind waits twice as long: a disturbance that has passed does not slow the engine for the rest of its life, and a clock that moves once
// in thousands of steps does not hunt.
static void guard_set_rate(cwh_guard *g, double rate)
{
    g->rate = rate;
    g->ms_sum = 0;
    g->ms_n = 0;
    g->good = g->late = 0;
    g->ref_ms = 0;
}
void cwh_guard_init(cwh_guard *g, double rate)
{
    memset(g, 0, sizeof(*g));
    g->rate = g->rate_top = rate;
    g->probe_need = CWH_GUARD_RECOVER;
    g->recover_need = CWH_GUARD_RECOVER;
}
int cwh_guard_step(cwh_guard *g, double ms, double scheduled)
{
    // the yardstick: the schedule, or what this rate delivered when a trial accepted it.  While a trial runs only sweeps FAR off -- 15 % over the
    // schedule AND over what the rate it left delivered -- end it early: its verdict is the mean.
    const double ref = std::max(scheduled, g->probing ? g->prev_mean : g->ref_ms);
    const bool late = ms > (g->probing ? 1.15 : 1.06) * ref;
    g->late = late ? g->late + 1 : 0;
    // ("in a row" for the way up means MOSTLY: at the edge one launch in ten is late by itself, and 64 strictly in a row would never come)
    g->good = late ? std::max(0, g->good - 8) : g->good + 1;
    if (g->ms_n >= 128) { g->ms_sum *= 0.5; g->ms_n /= 2; }              // (the mean is of the last ~100 samples, not of the rate's whole past)
    g->ms_sum += ms;
    g->ms_n++;
    const bool verdict_due = g->probing && g->ms_n >= CWH_GUARD_PROBE_SAMPLES;
    const double mean = g->ms_sum / g->ms_n;
    if (g->probing && (g->late >= 3 || (verdict_due && mean >= 0.998 * g->prev_mean))) {      // ---- a trial that does not pay: undone
        const double ref_prev = g->ref_prev;
        guard_set_rate(g, g->rate - CWH_GUARD_NOTCH);
        g->ref_ms = ref_prev;
        int32_t &need = g->recovering ? g->recover_need : g->probe_need;  // the next attempt of its kind waits twice as long (no see-saw between two notches)
        need = std::min(2 * need, (int32_t)CWH_GUARD_NEED_MAX);
        g->probing = g->recovering = 0;
        return CWH_GUARD_TRIAL_UNDONE;
    }
    if (verdict_due) {                                                   // ---- a trial that PAYS: kept, and its mean is this rate's yardstick
        if (g->recovering) g->recover_need = CWH_GUARD_RECOVER;
        else { g->rate_top = g->rate; g->probe_need = CWH_GUARD_RECOVER / 4; }      // (the next notch is tried sooner)
        g->ref_ms = mean;
        g->probing = g->recovering = 0;
        return CWH_GUARD_TRIAL_KEPT;
    }
    if (!g->probing && g->late >= 3 && g->rate > CWH_GUARD_RATE_FLOOR + 0.1) {          // ---- not keeping its schedule: a notch down
        guard_set_rate(g, g->rate - CWH_GUARD_NOTCH);
        g->slowdowns++;
        return CWH_GUARD_SLOWDOWN;
    }
    if (!g->probing && g->ms_n >= CWH_GUARD_PROBE_SAMPLES && g->rate + 0.05 < CWH_GUARD_RATE_CEILING &&
        g->good >= (g->rate + 0.1 < g->rate_top ? g->recover_need : g->probe_need)) {
        // ---- a trial: one notch up -- back towards the best rate known after a slowdown, or beyond it (a probe)
        g->recovering = g->rate + 0.1 < g->rate_top;
        g->prev_mean = mean;
        g->ref_prev = g->ref_ms;
        guard_set_rate(g, std::min(g->rate + CWH_GUARD_NOTCH, CWH_GUARD_RATE_CEILING));
        g->probing = 1;
        if (!g->recovering) g->probes++;
        return CWH_GUARD_TRIAL_UP;
    }
    return CWH_GUARD_NONE;
}

// ------------------------------------------------------------------------------ the look-ahead refill period
int32_t cwh_la_adapt(int32_t period, int32_t period_max, uint64_t slow_delta, int32_t *quiet)
{
    if (slow_delta * 8ull > (uint64_t)period) {          // (what a refill launch costs: ~2 us per step at a period of 8 against 12 us per slow reset)
        *quiet = 0;
        if (period > CWH_LA_PERIOD_MIN) period = period / 2 < CWH_LA_PERIOD_MIN ? CWH_LA_PERIOD_MIN : period / 2;
    } else if (slow_delta * 32ull <= (uint64_t)period && ++*quiet >= CWH_LA_QUIET) {
        *quiet = 0;
        if (period < period_max) period = period * 2 > period_max ? period_max : period * 2;
    }
    return period;
}

}  // extern "C"
