// cw_host.h -- the HIP-FREE host logic of the engine: MT19937 state conversion between numpy's form and the engine's, the dense view of a slot
// record, the checkpoint blob's section sizes, the DLPack producer, and the DECISIONS of the sweep clock's guard.  Plain C++ with a C ABI
// (cwh_*): compiled into libcraftingworld.so by hipcc, and on its own by g++ under -fsanitize=address,undefined for the CPU test tier
// (make -C gym_craftingworld_amd/csrc host_asan -> libcw_host_asan.so; tests/test_host_logic.py, tests/test_sanitizers.py).
#pragma once
#include <stddef.h>
#include <stdint.h>

#define CWH_MT_N 624

#ifdef __cplusplus
extern "C" {
#endif

// ---- MT19937: numpy RandomState (key, pos)  <->  the engine's consume-and-replace form (cw_mt.h)
int cwh_mt_from_numpy(uint32_t *s, int pos);                       // in place; returns the engine index (pos mod 624)
void cwh_mt_to_numpy(const uint32_t *s, int idx, uint32_t *key);   // engine (s, idx) -> a numpy key whose stream from position idx is identical
void cwh_mt_untwist(uint32_t *key);                                // one generation back (the twist's inverse)
void cwh_mt_rewind(uint32_t *key, int32_t *pos, uint32_t n);       // numpy state -> the state n raw draws earlier
void cwh_mt_init_genrand(uint32_t *s, uint32_t seed);              // numpy RandomState(int)

// ---- DLPack producer (a malloc'ed, non-owning DLManagedTensor over engine memory; device_type 10 = kDLROCM)
void *cwh_dlpack_make(void *data, int device_id, int code, int bits, int ndim, const int64_t *shape);

// ---- a slot record (8 cell indices u16 + 8 4-bit codes) as the dense grid of cw_state_view: grid[ncell] cell codes (0 empty)
void cwh_slots_to_grid(const uint16_t *pos, uint32_t codes, int ncell, uint8_t *grid);

// ---- checkpoint blob: byte sizes of its sections, in file order, for n envs / fixed_init_state pool k / la_depth look-ahead records per env (0: none kept).
// Writes at most CWH_CKPT_SECTIONS sizes, returns how many; *total (may be null) = their sum (the blob is header + total).
#define CWH_CKPT_SECTIONS 22
int cwh_ckpt_section_bytes(int64_t n, int32_t k, int32_t la_depth, size_t *sizes, uint64_t *total);

// ---- the GUARD of the sweep's clock as a pure state machine (cw_engine.cpp: sweep_guard_tick feeds it one timed sweep at a time; nothing here
// touches HIP).  Rates in TB/s, times in ms.  DESIGN.md 4.3; the constants are the ones round 4/5 measured (profiles/r04_clock.txt, r05_experiments.txt).
typedef struct cwh_guard {
    double rate;            // the clock's current rate
    double rate_top;        // the best rate known to hold: cw_create's choice, raised by a probe that paid
    double ms_sum;          // sweep times sampled at the current rate (decayed: the last ~100)
    double prev_mean;       // their mean at the rate a running trial left
    double ref_ms;          // what the current rate delivered when a trial ACCEPTED it (its yardstick if that is more than its schedule)
    double ref_prev;        // ... the one of the rate a running trial left
    int32_t ms_n;
    int32_t late;           // samples late in a row
    int32_t good;           // samples on time, "mostly in a row" (a late one costs 8)
    int32_t slowdowns;      // moves down that were not the end of a trial
    int32_t probes;         // trials beyond rate_top started
    int32_t probe_need;     // samples on time before the next probe (doubles after one that did not pay, capped)
    int32_t recover_need;   // ... before the next step back towards rate_top after a slowdown
    int32_t probing;        // a TRIAL is running: one notch up, verdict after CWH_GUARD_PROBE_SAMPLES samples
    int32_t recovering;     // ... and it is a step back towards rate_top, not beyond it
} cwh_guard;

enum { CWH_GUARD_NONE = 0, CWH_GUARD_SLOWDOWN = 1, CWH_GUARD_TRIAL_UP = 2, CWH_GUARD_TRIAL_KEPT = 3, CWH_GUARD_TRIAL_UNDONE = 4 };
#define CWH_GUARD_RECOVER 64
#define CWH_GUARD_PROBE_SAMPLES 32
#define CWH_GUARD_NEED_MAX 2048
#define CWH_GUARD_RATE_FLOOR 5.0
#define CWH_GUARD_RATE_CEILING 7.7
#define CWH_GUARD_NOTCH 0.2

void cwh_guard_init(cwh_guard *g, double rate);
// One timed sweep at the CURRENT rate: `ms` measured, `scheduled_ms` what its clock promises (cwh_guard_scheduled_ms).  Returns what the guard
// does about it (CWH_GUARD_*); on every action but NONE and TRIAL_KEPT g->rate has changed and the caller re-programs the clock.
int cwh_guard_step(cwh_guard *g, double ms, double scheduled_ms);
// The three periods of the clock at a rate, in 1/16 of a 10-ns tick (0 at rate 0 = unclocked): a wave starts a 4-KiB piece every period; a launch's
// first CWH_HEAD_JOBS jobs run head_notch TB/s slower, those after a step on which envs finished busy_notch slower (never under the floor).
#define CWH_HEAD_JOBS 64
void cwh_sweep_periods(double rate, int32_t sweep_waves, double head_notch, double busy_notch, int32_t *period16, int32_t *period16_head, int32_t *period16_busy);
// What a sweep of `sweep_jobs` jobs per wave should take with those periods, after a busy step, plus what a launch costs beside its jobs
double cwh_guard_scheduled_ms(double sweep_jobs, int32_t period16, int32_t period16_busy, double beside_ms);

// ---- the look-ahead refill period follows the episodes (cw_engine.cpp: la_adapt).  `slow_delta` slow-path resets were counted since the last refill was
// enqueued (read from a pinned word the refill kernels write; stale by a period, never waited for): more than an eighth of the period's steps -> half the
// period (CWH_LA_PERIOD_MIN at least); at most a 32nd of them for CWH_LA_QUIET refills in a row -> twice the period (period_max at most).  -> the new period.
#define CWH_LA_PERIOD_MIN 8
#define CWH_LA_QUIET 8
int32_t cwh_la_adapt(int32_t period, int32_t period_max, uint64_t slow_delta, int32_t *quiet);

#ifdef __cplusplus
}
#endif
