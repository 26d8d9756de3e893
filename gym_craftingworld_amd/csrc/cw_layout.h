// cw_layout.h -- HBM data layout of the batched CraftingWorld engine (host + device).
//
// State is SPARSE, not a dense grid: the reference world always holds exactly the 8 objects
// sample_state() placed (ray.py:599-628) -- they transform in place (tree->sticks, sticks->house,
// wheat->bread), vanish (rock, bread) or are carried, but never multiply -- so an env is
//   header (16 B) + 8 object slots (cell index u16 each, 4-bit code each).
// Every per-env array is a structure-of-arrays of 16-byte records, so one wavefront reads
// 64 x 16 B = 1 KiB contiguous per load (env i handled by lane i): fully coalesced, and the
// step kernel never makes a data-dependent (gather) access to HBM.  "What is in cell p" is 8
// register compares.  Dense views (grid codes, one-hot, pixels) are materialised by the
// render/export kernels from the slots.
#pragma once
#include <stdint.h>

#define CW_POS_GONE 0xFFFFu   // slot's object no longer exists
#define CW_POS_HELD 0xFFFEu   // slot's object is in the agent's hand

// hdr.flags
#define CW_FLAG_RESET 0x1u    // env was (re)set by the last cw_reset / auto-reset (render: also write desired/init frames)
#define CW_FLAG_SUBSET 0x2u   // reward_style is not None (compute_reward_subset, ray.py:763-767)
//   flags bits 2-15: steps of the running episode that returned MAX_STEPS (saturating at 16 383): the episode's return so far is a closed form of this count
//   and step_num (episode_return_of) -- exact also where an env WITHOUT auto-reset is stepped on after its first done, as the reference allows (ray.py:367)

// Packed header, one uint4 per env:
//   x = agent_r | agent_c << 8 | hold << 16 | menu << 24
//   y = achieved | desired << 16
//   z = step_num | flags << 16
//   w = slot codes, 4 bits each (slot k = bits 4k..4k+3); code 0 = gone
// Slot positions, one uint4 per env: 8 x u16, slot k = (word[k>>1] >> 16*(k&1)) & 0xFFFF.
// Slot k initially holds object k (code k+1) -- OBJECTS order, ray.py:21.

#define CW_CODES_INITIAL 0x87654321u  // slot k has code k+1
#define CW_CTL_QUEUED 0x100u          // nx_ctl: the env's ring wants a refill (a record was taken, or it was reset the slow way): the refill kernel scans for this bit; bits 0-7: the queue's head slot
#define CW_LA_DEPTH 4                 // look-ahead records kept per env (round 6: a QUEUE -- an env that finishes two, three, four times between two refills
                                      // still finds a record; rounds 4-5 kept one, and every further finish was reset the slow way on the step's critical path)

// One ordered selected_tasks list, device form (cw_task_menu packed): selected_bits as nibbles.
struct CwMenuDev {
    uint64_t sel_bits;       // nibble i = task_list.index(selected_tasks[i])
    int32_t n_selected;
    int32_t number_of_tasks;
    int32_t stacking;
    int32_t reward_subset;
};

// Host-side launch tuning of one engine (defaults = the measured best; the CW_TUNE_* environment variables read in
// cw_create override them for experiments).
struct CwTuning {
    int n_cu = 256;                 // compute units of the engine's device (hipDeviceProp_t::multiProcessorCount, set by cw_create)
    int period16 = 0;               // the sweep's CLOCK: a wave's jobs (4-KiB pieces) start one period apart; in 1/16 of a 10-ns tick of the 100-MHz clock (0: unclocked)
    int period16_head = 0;          // ... of a launch's first 64 jobs (the write path takes a notch less from a launch's first ~40 us) ...
    int period16_busy = 0;          // ... and of those after a step on which envs finished (their frames were written just before the sweep)
    int render_chunk_rounds = 896;  // a large batch is swept in launches of at most this many rounds of 3 KB per wave over consecutive env ranges:
                                    // 131 072 envs at 21x21 (0: one launch whatever the batch)
    int gather = 1;                 // the smallest Ray frames (grids up to gather_max_size) by cw_render_gather_kernel; 0: by the piece sweep like all others (CW_TUNE_GATHER)
    int gather_max_size = 7;        // (measured: the gather painter wins up to 7x7, the piece sweep from 8x8 on; profiles/r05_small_frames.txt)
    int small_frame_bytes = 4096;   // frames under this many bytes are swept with small_blocks_per_cu workgroups per CU instead of one (CW_TUNE_SMALL_FRAME_BYTES)
    int small_blocks_per_cu = 4;    // (CW_TUNE_SMALL_BLOCKS)
    long long small_launch_bytes = 320ll << 20;   // ... the piece sweep only while a launch writes less than this (the gather painter always); CW_TUNE_SMALL_LAUNCH_MB
    int step_envs_per_wave = 64;    // most envs a wave of cw_step_fused_kernel steps (64: one wave per SIMD at 65 536 envs; CW_TUNE_STEP_ENVS_PER_WAVE: 8 / 16 / 32 / 64)
    int reset_blocks_per_cu = 4;    // resetting workgroups (4 waves = 4 envs in flight each) per CU at most: the reset, refill and pool kernels (round 6: 2 -> 4;
                                    // nothing runs beside the sweep any more, the refill's scan wants the waves: CW_TUNE_RESET_BLOCKS, profiles/r06_experiments.txt D)
};

// Control block of the RESIDENT stepper (cw_step_resident: the single-env loop, a step without a kernel launch): pinned, coherent host memory
// the device polls.  One 128-byte line per direction so that the host's doorbell stores and the device's answers never share a line.
struct CwResident {
    unsigned long long bell; // host -> device, ONE word so that a poll is one PCIe read: low half (seq << 8) | action, seq = 1, 2, ... (a new seq is a
                             // new step request); high half 1 = leave now (cw_resident_stop and every entry point that touches the engine's state)
    uint32_t pad0[30];
    uint32_t ack;            // device -> host: the last seq whose outputs (reward, done, masks, repainted cells, state) are visible
    uint32_t exited;         // device -> host: 0 while resident; else reason (1 stop, 2 idle, 3 time slice used up) | last seq served << 8
    uint32_t pad1[30];
};

// Everything the kernels need, passed by value.
struct CwParams {
    // per-env state (SoA of 16-byte records unless noted)
    uint4 *hdr;              // [N]
    uint4 *pos;              // [N] current slot positions
    uint4 *init_pos;         // [N] positions at reset of objects 0..7 (INIT_OBS_VECTOR, ray.py:183)
    uint16_t *init_agent;    // [N] agent cell at reset
    uint4 *goal_pos;         // [N] imagine_obs final_state slot positions (ray.py:220-299)
    uint32_t *goal_codes;    // [N]
    uint16_t *goal_agent;    // [N]
    int32_t *ep_no;          // [N]
    // MT19937 per env, env-contiguous [N][624], "consume-and-replace" convention (cw_mt.h)
    uint32_t *mt;
    int32_t *mt_idx;         // [N] next word index 0..623
    // fixed_init_state pool: [N][K][9] u16 (objects 0..7 + agent cell)
    uint16_t *pool;
    // outputs
    int32_t *reward;         // [N]
    uint8_t *done;           // [N]
    uint16_t *achieved_out;  // [N]
    uint16_t *desired_out;   // [N]
    int32_t *episode_length; // [N]
    int32_t *episode_return; // [N] sum of the finished episode's rewards (ray.py:361-367), written where done
    uint8_t *obs;            // [N][P][P][3] or null
    uint8_t *desired_img;
    uint8_t *init_img;
    uint8_t *terminal_img;   // or null
    // LOOK-AHEAD: the outcome of every env's NEXT reset(), computed ahead of time.  Only reset() draws from an env's RNG stream, so the next
    // episode's placement, goal state and task set are known as soon as the previous reset has been taken: the refill kernel runs them ahead in
    // bulk, off the per-step path, and a finished env just takes the record over inside the step kernel (a POP: three 16-byte loads).  An env
    // that finishes again before the next refill finds no record and is reset the slow way, on the spot, from the same stream position.
    // The records of an env form a RING of CW_LA_DEPTH slots, slot d of env e at [d * N + e] (one coalesced array per slot): nx_ctl[e] names the head slot --
    // the NEXT episode --, the valid records follow it in stream order, a pop invalidates the head and advances it (one round trip: the three 16-byte loads of
    // the head slot, as with the single record of rounds 4-5), the refill computes the missing ones behind the last valid one.
    uint4 *nx_init_pos;      // [D][N] sample_state placement of the episode
    uint4 *nx_goal_pos;      // [D][N] its imagine_obs final state
    uint4 *nx_misc;          // [D][N] x = init_agent | goal_agent << 16, y = goal_codes, z = desired | subset << 16 | VALID << 31, w = raw 32-bit draws the
                             //     record consumed (cw_get_mt rewinds the exported stream by the draws of every waiting record)
    uint32_t *nx_ctl;        // [N] head slot (bits 0-7) | CW_CTL_QUEUED; read with the env's state when a step begins
    int32_t lookahead;       // 0: no records are kept (engines without auto-reset, host-mapped engines, CW_TUNE_LOOKAHEAD=0)
    unsigned long long *la_feedback;   // pinned host word or null: every refill kernel leaves counters[5] (resets taken the slow way so far) here -- the host
                                       // adapts its refill period to it without ever waiting for the card (cw_engine.cpp: cw_step)
    unsigned long long *counters; // [4] public: steps, finished, successes, invalid actions; [4] PRIVATE: the finished count the last sweep of the
                                  // observation array saw (cw_render_pieces_kernel: what kind of step does it follow?), [5] PRIVATE: resets of
                                  // look-ahead engines that found no record and were taken the slow way; 8 words allocated
    const CwMenuDev *menus;
    // constants
    int32_t n_envs;
    int32_t size;            // S
    int32_t ncell;           // S*S
    int32_t max_steps;
    uint32_t task_mask;      // (1 << n_task_list) - 1
    int32_t pool_k;          // fixed_init_state
    uint32_t div_magic;      // floor(2^32 / S) + 1 : x / S == mulhi(x, magic) for x < 2^18
    uint32_t frame_bytes;    // 48 * S * S, or 27 * S * (S+1) for the AltObs rasteriser
    int32_t raster;          // CW_RASTER_*
    uint8_t *res_onehot;     // resident stepper only (else null): host-mapped [S][S][12] one-hot state, rewritten after every resident step
};
