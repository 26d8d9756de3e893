/*
 * cw_oracle.h -- CPU ORACLE for the CraftingWorld step/reset hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the reference's algorithm
 * (lauradarcy/gym-craftingworld, gym_craftingworld/envs/craftingworld_ray.py, "ray.py" below)
 * used as the parity checker by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 * Nothing under gym_craftingworld_amd/ (the product) may include, link or call it.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks this oracle bit-for-bit against
 * the .npz fixtures under tests/golden/, which tools/gen_golden.py captured by importing and running the reference
 * itself in the build container (seeded at the MT19937-state level).  The
 * seed(int) -> MT key hashing of gym<=0.21 is outside the pinned path ("parity unpinned" for
 * that mapping only; it is a host-side convenience in the product's seeding.py).
 *
 * Representation: one code per cell, 0 = empty, k+1 = OBJECTS[k] (ray.py:21)
 *   1 sticks 2 axe 3 hammer 4 rock 5 tree 6 bread 7 house 8 wheat
 * agent (row, col) and hold (0 none, 1 sticks, 2 axe, 3 hammer) are kept beside the grid; this is
 * the reference's (H,W,12) one-hot with channels 0-7 / 8 / 9-11 folded (at most one object per
 * cell is a reference invariant: drop needs an empty cell, ray.py:334).
 */
#ifndef CW_ORACLE_H
#define CW_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CWO_MT_N 624
#define CWO_MAX_TASKS 16

typedef struct cwo_config {
    int32_t size;              /* STATE_W == STATE_H (non-square is a reference defect, rejected) */
    int32_t max_steps;         /* ray.py:76 */
    int32_t reward_subset;     /* reward_style is not None -> compute_reward_subset, ray.py:71-74 */
    int32_t stacking;          /* ray.py:83 */
    int32_t n_task_list;       /* len(task_list), 9..16 */
    int32_t n_selected;        /* len(selected_tasks) */
    int32_t number_of_tasks;   /* already clamped to n_selected, ray.py:79-81 */
    int32_t fixed_init_state;  /* 0 = sample a fresh placement every reset, ray.py:116-118 */
    int32_t selected_bits[CWO_MAX_TASKS]; /* task_list.index(selected_tasks[i]), ray.py:174 */
    int32_t alt_obs;           /* 1: CraftingWorldEnvAltObs rasteriser (craftingworld_altobs.py:489-642): 3x3 px per cell,
                                * images are [(3*size+3)][3*size][3] */
} cwo_config;

typedef struct cwo_env cwo_env;

/* read-only view of one env (pointers stay valid for the env's lifetime) */
typedef struct cwo_view {
    const uint8_t *grid;        /* [size*size] codes, row-major              (obs_one_hot[:,:,:8]) */
    const uint8_t *init_grid;   /* codes at reset                             (INIT_OBS_VECTOR)    */
    const uint8_t *goal_grid;   /* imagine_obs final_state codes                                   */
    const uint8_t *obs;         /* [4*size][4*size][3] uint8                  (obs_image)          */
    const uint8_t *desired_img; /* desired_goal image                                              */
    const uint8_t *init_img;    /* INIT_OBS image                                                  */
    int32_t agent_r, agent_c, hold;
    int32_t goal_agent_r, goal_agent_c;
    int32_t init_agent_r, init_agent_c;
    uint32_t achieved, desired; /* bit i = task_list[i] */
    int32_t step_num, ep_no;
} cwo_view;

cwo_env *cwo_new(const cwo_config *cfg);
void cwo_free(cwo_env *e);

/* RNG: numpy RandomState state (key[624], pos) -- MT19937, legacy randint/shuffle */
void cwo_set_rng(cwo_env *e, const uint32_t *key, int32_t pos);
void cwo_get_rng(const cwo_env *e, uint32_t *key, int32_t *pos);
void cwo_seed_int(cwo_env *e, uint32_t seed); /* == numpy RandomState(seed): init_genrand */
uint32_t cwo_rng_u32(cwo_env *e);             /* raw genrand_uint32 (tests)   */
uint32_t cwo_rng_randint(cwo_env *e, uint32_t n); /* RandomState.randint(n)   */
void cwo_rng_shuffle(cwo_env *e, int32_t *x, int32_t n); /* RandomState.shuffle(arange) */

void cwo_generate_fixed_states(cwo_env *e);  /* ray.py:149-154, draws from the env RNG */
void cwo_get_fixed_states(const cwo_env *e, uint16_t *out /* [K][9]: cells of objects 0..7 and of the agent */);
void cwo_reset(cwo_env *e);                  /* ray.py:156-218 */
/* ray.py:301-378; returns 0, or -1 for an action outside [0,6) (state untouched) */
int cwo_step(cwo_env *e, int32_t action, int32_t *reward, int32_t *done);
void cwo_get_view(const cwo_env *e, cwo_view *v);
/* overwrite the dynamic state (parity injection); grids are size*size codes */
void cwo_set_state(cwo_env *e, const uint8_t *grid, const uint8_t *init_grid, int32_t agent_r,
                   int32_t agent_c, int32_t hold, uint32_t achieved, uint32_t desired,
                   int32_t step_num);
/* craftingworld_altobs.py:489-595 render(state): out[(3s+3)*3s*3], values modulo 256 (the reference's int
 * image reaches 2 x colour when the agent holds sticks on a sticks cell) */
void cwo_render_alt(int32_t size, const uint8_t *grid, int32_t agent_r, int32_t agent_c, int32_t hold,
                    uint8_t *out);
/* ray.py:442-520 render(state): full frame from (grid, agent, hold) into out[4s*4s*3] */
void cwo_render(int32_t size, const uint8_t *grid, int32_t agent_r, int32_t agent_c, int32_t hold,
                uint8_t *out);

/* Batched driver used as bench.py's cpu_baseline ("port"): T steps over n envs with auto-reset,
 * actions[t*n + i]; split over nthreads OpenMP threads.  rewards/dones may be NULL.
 * Returns the number of env-steps executed. */
int64_t cwo_batch_rollout(cwo_env **envs, int32_t n, const int8_t *actions, int32_t T,
                          int32_t nthreads, int32_t *rewards, uint8_t *dones);
int64_t cwo_batch_rollout_full(cwo_env **envs, int32_t n, const int8_t *actions, int32_t T, int32_t nthreads);

#ifdef __cplusplus
}
#endif
#endif
