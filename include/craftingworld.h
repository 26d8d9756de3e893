/*
 * craftingworld.h -- C ABI of the MI355X-native batched CraftingWorld step/reset engine.
 *
 * The reference (lauradarcy/gym-craftingworld) has no FFI/plugin interface: its boundary is the
 * gym Python API of CraftingWorldEnvRay (gym_craftingworld/envs/craftingworld_ray.py, "ray.py").
 * This header is the boundary a binding for that API calls into; every entry point cites the
 * reference interface it replaces.  Plain C: pointers, sizes, integer status codes.  No torch or
 * HIP types appear in the signatures (cw_stream_t is a hipStream_t passed as void*).
 *
 * Threading: one engine per device; calls on one engine are serialized by the caller.  Every
 * call that takes a stream only ENQUEUES work on it (no host synchronisation) unless stated.
 * Ownership: the engine owns all device buffers for its lifetime; cw_buffers() exposes them for
 * zero-copy wrapping (valid until cw_destroy; contents valid until the next cw_step/cw_reset --
 * the same live-alias contract as the reference, whose step() returns obs_image itself).
 */
#ifndef CRAFTINGWORLD_H
#define CRAFTINGWORLD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CW_ABI_VERSION 5   /* 5: cw_buffer_table.episode_return, cw_get_fixed_states; 4: cw_tuner_state, one painter, look-ahead records.  Round 6 changed no
                            * signature or struct: cw_get_mt reports numpy's own (key, pos) form, cw_rollout issues one launch per max_steps steps, checkpoint blobs
                            * are version 4 (a ring of look-ahead records per env; older blobs are refused with CW_ERR_INVALID), hdr flags bits 2-15 count successes */
#define CW_MT_N 624        /* MT19937 words per env (numpy RandomState key)        */
#define CW_MAX_TASKS 16    /* len(task_list) upper bound (bits of the goal masks)  */
#define CW_MAX_MENUS 256   /* distinct ordered selected_tasks lists per engine     */
#define CW_NUM_OBJECTS 8   /* OBJECTS, ray.py:21                                   */

/* status codes */
#define CW_OK 0
#define CW_ERR_INVALID (-1)   /* bad argument / config (cw_last_error() has the text) */
#define CW_ERR_HIP (-2)       /* a HIP runtime call failed                             */
#define CW_ERR_STATE (-3)     /* call order (e.g. cw_step before cw_reset)             */

/* cell codes of the dense views: 0 empty, k+1 = OBJECTS[k] (ray.py:21) */
enum { CW_EMPTY = 0, CW_STICKS = 1, CW_AXE = 2, CW_HAMMER = 3, CW_ROCK = 4, CW_TREE = 5,
       CW_BREAD = 6, CW_HOUSE = 7, CW_WHEAT = 8 };

/* observation modes (cw_config.obs_mode) */
enum {
    CW_OBS_STATE = 0,        /* no pixel buffers; state tensors only (BASELINE config 2)            */
    CW_OBS_PIXELS_FULL = 1,  /* render(): whole (4S,4S,3) uint8 frame rewritten every step, ray.py:442-520 */
    CW_OBS_PIXELS_DIRTY = 2  /* render_edit(): persistent frame, <=2 changed cells repainted, ray.py:522-557 */
};

/* rasterisers (cw_config.raster) */
enum {
    CW_RASTER_RAY = 0,  /* CraftingWorldEnvRay: 4x4 px per cell in the object's colour, frames [4S][4S][3]   (ray.py:442-557) */
    CW_RASTER_ALT = 1   /* CraftingWorldEnvAltObs: 3x3 px per cell, pixel k lit with CPV_COLORS[k] iff item k is in the
                         * cell, + a 3-px strip with a "holding" flag: frames [3S+3][3S][3] (craftingworld_altobs.py:489-642);
                         * values are the reference's int image modulo 256 */
};

/* action dtypes accepted by cw_step */
enum { CW_ACT_I32 = 0, CW_ACT_I64 = 1, CW_ACT_U8 = 2 };

/* One ordered selected_tasks list with its draw rules -- the ctor kwargs selected_tasks,
 * number_of_tasks, stacking, reward_style of ray.py:59-83.  Kept as an ORDERED list because
 * reset() shuffles indices into it (ray.py:171-174). */
typedef struct cw_task_menu {
    int32_t n_selected;                  /* len(selected_tasks), 1..CW_MAX_TASKS                   */
    int32_t number_of_tasks;             /* ray.py:79-81 (clamped to n_selected by cw_create)      */
    int32_t stacking;                    /* `stacking is True`, ray.py:169                         */
    int32_t reward_subset;               /* reward_style is not None -> compute_reward_subset      */
    int32_t selected_bits[CW_MAX_TASKS]; /* task_list.index(selected_tasks[i]), ray.py:174         */
} cw_task_menu;

/* CraftingWorldEnvRay.__init__ kwargs (ray.py:59-60) for a batch of num_envs envs */
typedef struct cw_config {
    int32_t abi_version;        /* CW_ABI_VERSION */
    int32_t num_envs;           /* N */
    int32_t size;               /* STATE_W == STATE_H, 4..255 (non-square rejected: reference defect) */
    int32_t max_steps;          /* MAX_STEPS, 1..65535 */
    int32_t n_task_list;        /* len(task_list), 9..CW_MAX_TASKS */
    int32_t fixed_init_state;   /* 0, or pool size K per env (ray.py:116-118), K <= 64 */
    int32_t obs_mode;           /* CW_OBS_* */
    int32_t auto_reset;         /* 1: cw_step resets finished envs itself (gym.vector semantics);
                                 * 0: finished envs keep stepping until cw_reset (single gym.Env semantics, ray.py:367) */
    int32_t keep_terminal_obs;  /* 1 (pixel modes + auto_reset): before a finished env is reset, its last frame is
                                 * painted into cw_buffer_table.terminal_obs (gym.vector's info["terminal_observation"]) */
    int32_t raster;             /* CW_RASTER_* (pixel modes) */
    int32_t host_outputs;       /* 1: the step outputs and the frames (every cw_buffer_table pointer down to episode_length) live in
                                 * pinned, GPU-mapped HOST memory and are valid on the host too; cw_buffer_table.host_actions is a
                                 * mapped int32[N] to write actions into.  For the single-env gym loop (ray.py step()/reset() called
                                 * from host code, docs/source/envs/gen_info.rst:62-82): a step is one launch + cw_synchronize, no
                                 * copies.  Kernel stores then cross PCIe -- not for large batches. */
    int32_t n_menus;            /* 1..CW_MAX_MENUS */
    const cw_task_menu *menus;  /* host array [n_menus] */
    const uint8_t *env_menu;    /* host array [num_envs] of menu ids, or NULL (= all envs use menu 0) */
} cw_config;

typedef struct cw_engine cw_engine;
typedef void *cw_stream_t;      /* hipStream_t */

/* Device buffers owned by the engine (cw_buffers).  N = num_envs, S = size, P = 4*S.
 * Pixel buffers are NULL in CW_OBS_STATE. */
typedef struct cw_buffer_table {
    uint8_t *obs;            /* [N][P][P][3]  observation == achieved_goal image (ray.py:194-196) */
    uint8_t *desired_goal;   /* [N][P][P][3]  imagine_obs() image, rewritten at reset (ray.py:191) */
    uint8_t *init_obs;       /* [N][P][P][3]  INIT_OBS, rewritten at reset (ray.py:193)            */
    uint8_t *terminal_obs;   /* [N][P][P][3]  last frame of the episode that just ended, valid where done==1 (NULL unless keep_terminal_obs) */
    int32_t *reward;         /* [N]  -1 or max_steps (ray.py:361-363)                              */
    uint8_t *done;           /* [N]  0/1 (ray.py:367); done envs have already been auto-reset      */
    uint16_t *achieved;      /* [N]  achieved_goal_vector as a bit mask AFTER the step, BEFORE auto-reset */
    uint16_t *desired;       /* [N]  desired_goal_vector of the episode the step belonged to       */
    int32_t *episode_length; /* [N]  step_num at done (valid where done==1)                        */
    int32_t *episode_return; /* [N]  sum of the episode's rewards up to a step that returned done==1 (valid where done==1), as the reference's
                              *      loop sums them (ray.py:361-367: -1 per step, MAX_STEPS on a step that leaves the goal satisfied).  With
                              *      auto_reset the episode ends there: max_steps - (length - 1) after a success, -length after a time-out.
                              *      WITHOUT auto_reset a finished env keeps stepping until cw_reset (ray.py:367) and every later done step
                              *      rewrites episode_length / episode_return with the sums up to that step, repeated success rewards included.
                              *      cw_rollout writes both at EVERY done step of its n_steps, like n_steps calls of cw_step. */
    uint8_t *hdr;            /* [N][16] packed per-env header of the CURRENT state (after auto-reset):
                              *   byte 0 agent row, 1 agent col, 2 hold (0 none,1 sticks,2 axe,3 hammer), 3 menu id,
                              *   bytes 4-5 achieved mask (LE u16), 6-7 desired mask, 8-9 step_num, 10-11 flags (bit 0: no step
                              *   taken yet in this episode, bit 1: subset reward rule, bits 2-15: steps of this episode that returned max_steps),
                              *   bytes 12-15 the 8 object slots' codes, 4 bits each (slot k in bits 4k..4k+3)      */
    uint16_t *slot_pos;      /* [N][8] cell index (row*S+col) of object slot k; 0xFFFF gone, 0xFFFE held          */
    uint64_t *counters;      /* [4]  {env-steps, episodes finished, successes (reward==max_steps), invalid actions}; the allocation holds
                              *      8 words: [4] is the engine's own (the finished count the last sweep of the observation array saw --
                              *      the sweep paces its first jobs by what the step before it did), [5] counts resets of a look-ahead engine that
                              *      found no record waiting (performance diagnostics), [6..7] unused.  Read-only for callers. */
    size_t frame_bytes;      /* P*P*3 (CW_RASTER_RAY) or (3S+3)*3S*3 (CW_RASTER_ALT) */
    int32_t *host_actions;   /* [N]  cw_config.host_outputs only (else NULL): mapped host buffer usable as cw_step's actions (CW_ACT_I32) */
    uint8_t *host_onehot;    /* [S][S][12] engines that can run cw_step_resident only (else NULL): obs_one_hot (ray.py:119) of the env in pinned host
                              * memory, rewritten by every cw_step_resident (CraftingWorldEnvOneHot returns it as the observation, onehot.py:369-371) */
} cw_buffer_table;

/* Host-side dense snapshot for parity injection / checkpointing (cw_get_state, cw_set_state).
 * All pointers are HOST arrays supplied by the caller; a NULL pointer skips that field. */
typedef struct cw_state_view {
    uint8_t *grid;        /* [N][S][S] cell codes            (obs_one_hot[:,:,:8], ray.py:119)   */
    uint8_t *init_grid;   /* [N][S][S] codes at reset        (INIT_OBS_VECTOR, ray.py:183)       */
    uint8_t *goal_grid;   /* [N][S][S] imagine_obs final_state codes (set: restores a checkpoint's goal)  */
    uint8_t *agent_rc;    /* [N][2]                          (agent_pos)                          */
    uint8_t *init_agent_rc; /* [N][2] agent cell at reset (channel 8 of INIT_OBS_VECTOR)              */
    uint8_t *goal_agent_rc; /* [N][2] agent cell of the goal state                                    */
    uint8_t *hold;        /* [N]                                                                   */
    uint16_t *achieved;   /* [N]                             (achieved_goal_vector)               */
    uint16_t *desired;    /* [N]                             (desired_goal_vector)                */
    int32_t *step_num;    /* [N]                                                                   */
    int32_t *ep_no;       /* [N]                                                                   */
} cw_state_view;

/* --- lifetime: replaces CraftingWorldEnvRay.__init__ (ray.py:59-143) for N envs ------------- */
int cw_create(const cw_config *cfg, int device, cw_engine **out);
int cw_destroy(cw_engine *e);

/* --- RNG: replaces seed() (ray.py:145-147) ----------------------------------------------------
 * cw_seed_mt injects numpy RandomState states: keys[N][624], pos[N] (RandomState.get_state()[1:3]).
 * cw_seed_int seeds env i like numpy RandomState(seeds[i]) (init_genrand).  Both are synchronous
 * host calls; the conversion itself runs on the device, one lane per env.  cw_get_mt returns states a numpy RandomState accepts via set_state and that
 * continue the identical stream, in numpy's own form: pos in 1..624 (a stream at a generation's end is reported as (that generation, 624), as
 * RandomState.get_state() does, not as (the next generation, 0)) -- after at least one draw the (key, pos) pair EQUALS the reference generator's.
 * These and the other synchronous entry points (cw_get_state, cw_set_state, cw_get_fixed_states, cw_checkpoint_*) wait for THIS ENGINE'S
 * work only -- whatever it enqueued on the streams it was handed since the last wait -- and do their copies on a stream of the engine's
 * own: another engine on the same device, or a learner, is not stalled (no device-wide synchronisation).  A stream handed to an enqueueing
 * call should stay alive until cw_synchronize on it (or one of these calls) has returned; if it was destroyed earlier the engine falls
 * back to one device-wide wait.  Work the engine cannot know of -- replays of a HIP graph its calls were captured into run on whatever
 * stream the graph is launched on -- is the caller's to wait for (cw_synchronize on that stream) before a synchronous call. */
int cw_seed_mt(cw_engine *e, const uint32_t *keys, const int32_t *pos);
int cw_seed_int(cw_engine *e, const uint32_t *seeds);
int cw_get_mt(cw_engine *e, uint32_t *keys, int32_t *pos);

/* generate_fixed_states (ray.py:149-154): draw fixed_init_state placements per env from the
 * env's current RNG stream.  No-op when fixed_init_state == 0.  Returns after the pool is complete (one-time cost). */
int cw_generate_fixed_states(cw_engine *e, cw_stream_t stream);
/* fixed_state_list (ray.py:116-118): the pool as cell indices, HOST array out[N][K][9] uint16 = the cells (row*S+col) of objects 0..7
 * (OBJECTS order, ray.py:21) and of the agent in each of the K placements of every env.  Synchronous; CW_ERR_INVALID when K == 0. */
int cw_get_fixed_states(cw_engine *e, uint16_t *out);

/* --- reset() for every env (ray.py:156-218): task draw, placement, imagine_obs, render ------ */
int cw_reset(cw_engine *e, cw_stream_t stream);

/* --- step(action) for every env (ray.py:301-378) + auto-reset of finished envs --------------
 * actions: DEVICE pointer to N actions of dtype CW_ACT_*, values 0..5 = Up,Right,Down,Left,
 * PickUp,Drop (ACTIONS, ray.py:130-131).  Out-of-range values are counted in counters[3] and
 * executed as a state-preserving step (step_num += 1, reward -1).  Enqueues the step kernel -- engines with auto_reset: finished envs take
 * over the record of their next episode, computed ahead of time by a refill kernel that rides on every max_steps/4-th call (8 ... 64; fewer steps apart while
 * envs finish faster than that: the period follows the count of slow resets the card reports, never waited for); every env keeps a queue of four such records,
 * and one that finishes a fifth time between two refills is reset on the spot -- and, in CW_OBS_PIXELS_FULL, the sweep that paints the observation array.
 * With cw_config.host_outputs `actions` may be cw_buffer_table.host_actions. */
int cw_step(cw_engine *e, const void *actions, int action_dtype, cw_stream_t stream);

/* --- n_steps consecutive step()s from a DEVICE action array [n_steps][N] (dtype as cw_step): exactly what n_steps calls of cw_step enqueue,
 * in one call (a host loop in Python costs more per call than a state-only step takes on the card).  The frames and outputs left behind are
 * the last step's.  Capturable into a HIP graph as one piece (a replay re-reads the action array: refill it in place between replays).
 * A captured sequence carries ONE look-ahead refill at its head (plus the regular one every max_steps/4 steps inside it): a replayed graph must
 * refill by itself.  That launch costs ~15 us whatever it finds to do -- capture sequences of a refill period or more (a graph of a single
 * cw_step pays it on every replay: three times a 5-us state-only step).  An env that finds no record is reset on the spot and rejoins the list,
 * so a graph replayed after a re-seed has its records back after one episode. */
int cw_step_many(cw_engine *e, const void *actions, int action_dtype, int32_t n_steps, cw_stream_t stream);

/* --- n_steps consecutive step()s (+ auto-reset) for every env in ONE persistent kernel launch --
 * For scripted / random action streams known up front (BASELINE config 2 style): actions is a DEVICE
 * array [n_steps][N] of uint8 action ids; rewards [n_steps][N] int32 and dones [n_steps][N] uint8 are
 * DEVICE arrays or NULL.  The result (state, RNG streams, counters, the reward/done/achieved buffers of
 * the last step) is bit-identical to n_steps calls of cw_step.  CW_OBS_STATE engines with auto_reset
 * only: no frames are painted (use cw_render afterwards).  A call longer than max_steps steps is issued as several launches of max_steps steps
 * (64 at least) with a look-ahead refill ahead of each, in stream order: every env's time-out then finds its next episode's record waiting. */
int cw_rollout(cw_engine *e, const uint8_t *actions, int32_t n_steps, int32_t *rewards, uint8_t *dones,
               cw_stream_t stream);

/* step(action) of the SINGLE-ENV loop without a kernel launch (ray.py:301-378 called from host code, docs/source/envs/gen_info.rst:62-82).
 * For engines with num_envs == 1, host_outputs, auto_reset == 0 and obs_mode CW_OBS_STATE or CW_OBS_PIXELS_DIRTY: a resident
 * single-wavefront kernel polls a doorbell word in pinned host memory; this call rings it and spins until the step's outputs (reward, done,
 * masks, the <= 2 repainted cells of the host-mapped frame) are visible -- a few microseconds instead of a launch plus a stream
 * synchronisation.  Results are those of cw_step with the same action (0..127; larger ids than 5 are the counted no-op of cw_step).  The kernel is started on demand and leaves by itself: after 0.5 ms
 * without a request, after a 200-ms time slice, or when any other entry point of this engine is called (they park it first; cw_resident_stop
 * does only that).  Synchronous.  A new instance of the kernel waits for the stream of the engine's last cw_reset / cw_step / cw_rollout first, so
 * `cw_reset(e, s); cw_step_resident(e, a, 0);` needs no synchronisation in between. */
int cw_step_resident(cw_engine *e, int32_t action, int32_t want_onehot /* 1: also rewrite cw_buffer_table.host_onehot */);
int cw_resident_stop(cw_engine *e);

/* render(state=None) for every env into a caller-supplied DEVICE buffer [N][P][P][3] (works in
 * every obs_mode; ray.py:442-520).  Any pointer the rasteriser's stores accept (Ray: 4-byte aligned); a 16-byte
 * aligned one is painted by the fastest kernel. */
int cw_render(cw_engine *e, uint8_t *out_frames, cw_stream_t stream);

/* render(state) (ray.py:442-486) for caller-supplied one-hot states of ANY content: onehot is a DEVICE array [n_states][S][S][12]
 * uint8 (S = the engine's size), out_frames a DEVICE array [n_states][4S][4S][3] uint16 -- the reference's image is the SUM of the
 * colours of the objects in a cell (int; up to 8 x 255), the agent is the first cell (row-major) with channel 8 set and must exist,
 * the held item's colour comes from the largest hold channel set anywhere.  Does not touch the engine's own state.
 * Engines with CW_RASTER_ALT render CraftingWorldEnvAltObs.render(state) instead (craftingworld_altobs.py:489-560): out_frames is
 * [n_states][3S+3][3S][3] uint16, pixel k of a cell's tile = CPV_COLORS[k] x (channel k + hold channel 9+k for k < 3), the strip's
 * pixels 3..5 are 255 if any cell has a hold channel set. */
int cw_render_onehot(cw_engine *e, const uint8_t *onehot, int32_t n_states, uint16_t *out_frames, cw_stream_t stream);

/* Dense state views written to caller-supplied DEVICE buffers (observation_vector_space,
 * ray.py:94-110): cw_export_grid -> [N][S][S] uint8 codes; cw_export_onehot -> [N][S][S][12]
 * uint8 0/1 (channels 0-7 objects, 8 agent, 9-11 held item at the agent cell). */
int cw_export_grid(cw_engine *e, uint8_t *out, cw_stream_t stream);
int cw_export_onehot(cw_engine *e, uint8_t *out, cw_stream_t stream);
/* The same one-hot view of the episode's other two states -- what CraftingWorldEnvOneHot returns as desired_goal
 * (imagine_obs' final state, carftingworld_onehot.py:310) and init_observation (the state at reset, :203). */
enum { CW_STATE_CURRENT = 0, CW_STATE_GOAL = 1, CW_STATE_INIT = 2 };
int cw_export_onehot_of(cw_engine *e, int which, uint8_t *out, cw_stream_t stream);

/* --- state injection / checkpoint (synchronous; SURVEY §5 "checkpoint / resume") ------------ */
int cw_get_state(cw_engine *e, cw_state_view *host);
int cw_set_state(cw_engine *e, const cw_state_view *host);

/* --- checkpoint / resume of the whole batch as one opaque host blob (the reference has none: its de-facto
 * state is the attribute set of ray.py:119-141).  The blob holds the engine's raw records -- current state, the
 * episode's goal and start states, every env's RNG stream, the fixed_init_state pool, the last step's outputs, the
 * counters -- and is restored verbatim, so a resumed engine continues bit-identically, state tensors included.
 * cw_checkpoint_load needs an engine created with the same num_envs, size, max_steps, len(task_list),
 * fixed_init_state and task menus (verified; CW_ERR_INVALID otherwise); it repaints the frames in the pixel modes and
 * needs no cw_reset first.  Synchronous host calls. */
size_t cw_checkpoint_bytes(cw_engine *e);
int cw_checkpoint_save(cw_engine *e, void *buf, size_t capacity);
int cw_checkpoint_load(cw_engine *e, const void *buf, size_t length);

/* --- per-kernel timing with HIP events on the caller's stream (bench.py's roofline leg) -----
 * cw_profile_begin: from now on cw_step brackets its kernels with hipEventRecord on the stream it launches on (at most max_steps
 * steps are kept).  cw_profile_end: synchronises the events, returns durations in milliseconds and stops recording.
 * What is bracketed: in CW_OBS_PIXELS_FULL only the sweep of the observation array (cw_render_pieces_kernel; all chunk launches of a large
 * batch together) -- every event record costs the stream a pipeline bubble, and the step kernel in front of the sweep is short; in the
 * other two modes the step kernel (cw_step_fused_kernel / cw_step_kernel), which is the whole step there.  The look-ahead refill
 * (cw_refill_kernel, every max_steps/4-th step) is never bracketed. */
typedef struct cw_profile {
    int32_t steps;           /* cw_step calls recorded */
    float ms_step_kernel;    /* average per launch; 0 in CW_OBS_PIXELS_FULL (not bracketed there) */
    float ms_reset_kernel;   /* always 0 since ABI 4: no reset kernel runs inside a step (finished envs take look-ahead records in the step kernel) */
    float ms_render_kernel;  /* the sweep, average per step; 0 in CW_OBS_STATE and CW_OBS_PIXELS_DIRTY */
    float ms_render_kernel_max;
    float ms_render_kernel_min;
    float ms_render_kernel_median;
} cw_profile;
int cw_profile_begin(cw_engine *e, int max_steps);
int cw_profile_end(cw_engine *e, cw_profile *out);
/* Name of the kernel that paints this engine's frames, as a rocprofv3 kernel trace of the same run lists it (without template arguments):
 * CW_OBS_PIXELS_FULL: "cw_render_pieces_kernel" -- the one painter of whole frame arrays, a clocked sweep of aligned 4-KiB pieces; a trace
 * shows cw_render_pieces_kernel<raster, frames per job> (<0, 2> for Ray frames of 4 KiB and more, <1, 2> AltObs; 4 / 8 / 16 frames per job for
 * smaller frames) -- or "cw_render_gather_kernel" for the smallest Ray frames (grids up to 7x7; <4> / <8> frames per piece in a trace), where
 * every lane computes its own 16-byte chunks; this is what ms_render_kernel brackets.  CW_OBS_PIXELS_DIRTY: the step kernel itself, which repaints the <= 2 changed cells
 * ("cw_step_fused_kernel" with auto_reset, else "cw_step_kernel").  CW_OBS_STATE: "" (nothing is painted).  A static string. */
const char *cw_render_kernel_name(const cw_engine *e);

/* What the engine's tuning holds (full-frame mode): the period of the sweep's clock -- a wave starts a 4-KiB piece every
 * period16 / 16 ticks of the 100-MHz clock, 0: unclocked; period16_head: the period of a launch's first 64 jobs, period16_busy: of those after a
 * step on which envs finished --, and whether the engine
 * keeps look-ahead records (cw_config.auto_reset, device-resident outputs); `resident`: 1 if cw_step_resident can be used on this engine;
 * guard_slowdowns: how often the clock's guard has lowered the rate because sweeps stopped keeping their schedule (-1: no guard; the guard also
 * probes one notch UP when the best rate known has held for ~4 000 steps and keeps it if the sweeps get shorter: period16 may fall below cw_create's).
 * Only performance depends on any of it. */
typedef struct cw_tuner_state {
    int32_t period16, period16_head, period16_busy, lookahead, resident, guard_slowdowns;
} cw_tuner_state;
int cw_tuner(const cw_engine *e, cw_tuner_state *out);

int cw_buffers(cw_engine *e, cw_buffer_table *out);
/* Blocks the calling thread until everything enqueued on `stream` has finished (hipStreamSynchronize): the one host
 * synchronisation of the single-env loop, where step() returns Python scalars (ray.py:376-378). */
int cw_synchronize(cw_engine *e, cw_stream_t stream);
int cw_num_envs(const cw_engine *e);
int cw_abi_version(void);
const char *cw_last_error(void);   /* thread-local text of the last failing call */

#ifdef __cplusplus
}
#endif
#endif
