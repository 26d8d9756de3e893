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
// render/export kernels through an LDS-free compare chain against the slots.
#pragma once
#include <stdint.h>

#define CW_POS_GONE 0xFFFFu   // slot's object no longer exists
#define CW_POS_HELD 0xFFFEu   // slot's object is in the agent's hand

// hdr.flags
#define CW_FLAG_RESET 0x1u    // env was (re)set by the last cw_reset / auto-reset (render: also write desired/init frames)
#define CW_FLAG_SUBSET 0x2u   // reward_style is not None (compute_reward_subset, ray.py:763-767)

// Packed header, one uint4 per env:
//   x = agent_r | agent_c << 8 | hold << 16 | menu << 24
//   y = achieved | desired << 16
//   z = step_num | flags << 16
//   w = slot codes, 4 bits each (slot k = bits 4k..4k+3); code 0 = gone
// Slot positions, one uint4 per env: 8 x u16, slot k = (word[k>>1] >> 16*(k&1)) & 0xFFFF.
// Slot k initially holds object k (code k+1) -- OBJECTS order, ray.py:21.

#define CW_CODES_INITIAL 0x87654321u  // slot k has code k+1

// One ordered selected_tasks list, device form (cw_task_menu packed): selected_bits as nibbles.
struct CwMenuDev {
    uint64_t sel_bits;       // nibble i = task_list.index(selected_tasks[i])
    int32_t n_selected;
    int32_t number_of_tasks;
    int32_t stacking;
    int32_t reward_subset;
};

// Host-side launch tuning of one engine (defaults = the measured best; the CW_TUNE_* /
// CW_PROFILE_* environment variables read in cw_create override them for experiments).
struct CwTuning {
    int n_cu = 256;                 // compute units of the engine's device (hipDeviceProp_t::multiProcessorCount, set by cw_create)
    int render_blocks_per_cu = 1;   // render workgroups per CU (persistent, grid-stride over frames)
    int render_blocks_abs = 0;      // >0: absolute cap on render workgroups
    int render_threads = 256;       // threads per render workgroup (64, 128, 256)
    int list_blocks = 0;            // workgroups of the done-list (terminal-frame) render (0: one per CU)
    int overlap = 1;                // full-pixel step: reset (+ its frames) on the side stream beside the main render
    int render_q_all = 0;           // XCD-aware frame shares (cw_create calibrates): rounds painted by every wave ...
    int render_fast_parity = -1;    //   ... the rest by workgroups of this index parity only (-1: equal shares)
    int fused_step = 1;             // state / dirty-cell modes with auto-reset: step + reset (+ paint) in one launch
    int profile_side = 0;           // profiling brackets every kernel, not just the dominant render kernel
    int piece_sweep = 1;            // the per-step render as a sweep of aligned 4-KiB pieces (render_pieces); cw_create keeps it if it measures faster than the other painter
    int piece_pace = 0;             // ... eighths of an s_sleep(1) per 1-KiB store (Ray raster: unpaced, AltObs: 4, unless cw_create measures another pace 3 % faster)
    int render_linear = 1;          // full-frame render as a linear sweep (job = a run of whole grid rows); 0: frame per wave
    int render_chunk_rounds = 896;  // ... in launches of at most this many rounds per wave over consecutive env ranges: 131 072 envs at 21x21 (0: one launch whatever the batch)
    int render_place = 3;           // one-launch full-frame step: which of the eight placements of the sweep's batch loop to launch (cw_render_step_kernel<k>;
                                    // tuned online by cw_step, CW_TUNE_RENDER_PLACE=k forces one)
    int render_pace_fine = 0;       // ... bits 16-23 of render_pace: iterations of a one-s_nop loop before every job (a pace finer than s_sleep's 64 clocks)
    int reset_blocks_per_cu = 2;    // resetting workgroups (4 waves = 4 envs in flight each) per CU at most: the reset kernels
    int fused_reset_blocks_per_cu = 1;   // ... and the resetting tail of the one-launch full-frame step (cwk_launch_step)
    int fused_render = 1;           // FULL pixel step: render + auto-resets in ONE launch (cw_render_step_kernel) instead of two kernels on two streams
    int render_pace = 0;            // linear sweep: bits 0-7 idle s_sleep(1) (64 clocks) per pair of jobs, bit 8 one more inside every job,
                                    // bits 12-15 more per pair while envs are being reset beside the sweep (cw_create sets 0x2100; cw_step tunes bits 12-15)
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
    uint8_t *obs;            // [N][P][P][3] or null
    uint8_t *desired_img;
    uint8_t *init_img;
    uint8_t *terminal_img;   // or null
    // done-list compaction: done_count[0] = entries, [1] = release ticket (cw_kernels.hip)
    int32_t *done_list;      // [N]
    int32_t *done_count;     // [2]
    unsigned long long *render_stats;   // [2] calibration only (else null): busy time of even / odd render workgroups' waves
    unsigned long long *counters; // [4]
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
    int32_t tune_reset_prio; // s_setprio 3 for: 2 the render waves and the resets inlined in the fused / rollout kernels (default), 1 every resetting wave, 0 nobody
    uint8_t *res_onehot;     // resident stepper only (else null): host-mapped [S][S][12] one-hot state, rewritten after every resident step
    int32_t alt_pace;        // AltObs frame painter: s_sleep(1) (64 clocks) after each 1-KiB store of the zero fill (cw_create calibrates)
    int32_t grp_rows;        // linear render: grid rows per 64-lane group = floor(64 / S) (0: S > 64, frame-per-wave render only)
    int32_t grp_per_frame;   // linear render: groups per frame = ceil(S / grp_rows)
};
