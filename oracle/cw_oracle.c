/*
 * cw_oracle.c -- CPU ORACLE (test infrastructure only; see cw_oracle.h for the rules).
 *
 * Restates, function by function, the hot path of the reference env
 *   gym_craftingworld/envs/craftingworld_ray.py  ("ray.py")  +  envs/coordinates.py ("coord.py")
 * and the two numpy.random.RandomState methods it calls (numpy is a third-party dependency of
 * the reference, unpinned in requirements.txt:2; the legacy RandomState stream is frozen by
 * NumPy policy -- algorithm restated below from numpy/random/src/mt19937 + legacy-distributions
 * and checked against the installed numpy in tests/test_oracle_rng.py).
 *
 * Pinned against golden vectors captured from the reference itself: tests/test_oracle_golden.py.
 */
#include "cw_oracle.h"

#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

enum { EMPTY = 0, STICKS = 1, AXE = 2, HAMMER = 3, ROCK = 4, TREE = 5, BREAD = 6, HOUSE = 7, WHEAT = 8 };
/* TASK_LIST order, ray.py:40-41 */
enum { T_MAKEBREAD = 0, T_EATBREAD = 1, T_BUILDHOUSE = 2, T_CHOPTREE = 3, T_CHOPROCK = 4,
       T_GOTOHOUSE = 5, T_MOVEAXE = 6, T_MOVEHAMMER = 7, T_MOVESTICKS = 8 };

/* COLORS_N, ray.py:28-30 (index = cell code) */
static const uint8_t COLORS_N[9][3] = {
    {0, 0, 0},       {110, 69, 39},   {255, 105, 180}, {100, 100, 200}, {100, 100, 100},
    {0, 128, 0},     {205, 133, 63},  {197, 91, 97},   {240, 230, 140}};
/* COLORS_H, ray.py:31 (index = hold-1) */
static const uint8_t COLORS_H[3][3] = {{145, 186, 216}, {0, 150, 75}, {155, 155, 55}};

struct cwo_env {
    cwo_config cfg;
    int ncell;
    /* numpy RandomState: MT19937 key + pos */
    uint32_t mt[CWO_MT_N];
    int mti;
    /* dynamic state */
    uint8_t *grid, *init_grid, *goal_grid;
    uint8_t *obs, *desired_img, *init_img;
    int agent_r, agent_c, hold;
    int goal_agent_r, goal_agent_c;
    int init_agent_r, init_agent_c;
    uint32_t achieved, desired;
    int step_num, ep_no;
    /* fixed_init_state pool: K grids + agent positions */
    uint8_t *pool_grid;
    int *pool_agent;
};

/* ------------------------------------------------------------------ MT19937 (numpy mt19937.c) */
static void mt_init_genrand(cwo_env *e, uint32_t s)
{
    e->mt[0] = s;
    for (int i = 1; i < CWO_MT_N; i++)
        e->mt[i] = 1812433253u * (e->mt[i - 1] ^ (e->mt[i - 1] >> 30)) + (uint32_t)i;
    e->mti = CWO_MT_N;
}

static void mt_gen(cwo_env *e)
{
    const int N = 624, M = 397;
    uint32_t *mt = e->mt, y;
    int i;
    for (i = 0; i < N - M; i++) {
        y = (mt[i] & 0x80000000u) | (mt[i + 1] & 0x7fffffffu);
        mt[i] = mt[i + M] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    for (; i < N - 1; i++) {
        y = (mt[i] & 0x80000000u) | (mt[i + 1] & 0x7fffffffu);
        mt[i] = mt[i + (M - N)] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    y = (mt[N - 1] & 0x80000000u) | (mt[0] & 0x7fffffffu);
    mt[N - 1] = mt[M - 1] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    e->mti = 0;
}

uint32_t cwo_rng_u32(cwo_env *e)
{
    if (e->mti >= CWO_MT_N) mt_gen(e);
    uint32_t y = e->mt[e->mti++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

/* legacy random_interval(max): smallest all-ones mask >= max, draw u32 & mask until <= max;
 * max == 0 draws nothing (numpy legacy-distributions.c) */
static uint32_t rng_interval(cwo_env *e, uint32_t max)
{
    if (max == 0) return 0;
    uint32_t mask = max, v;
    mask |= mask >> 1;
    mask |= mask >> 2;
    mask |= mask >> 4;
    mask |= mask >> 8;
    mask |= mask >> 16;
    while ((v = (cwo_rng_u32(e) & mask)) > max) {}
    return v;
}

/* RandomState.randint(n) (low=0, high=n, dtype int64, masked rejection): interval(n-1) */
uint32_t cwo_rng_randint(cwo_env *e, uint32_t n) { return rng_interval(e, n - 1); }

/* RandomState.shuffle on a 1-d array: for i = n-1 .. 1: j = interval(i); swap(x[i], x[j]) */
void cwo_rng_shuffle(cwo_env *e, int32_t *x, int32_t n)
{
    for (int i = n - 1; i >= 1; i--) {
        uint32_t j = rng_interval(e, (uint32_t)i);
        int32_t t = x[i];
        x[i] = x[j];
        x[j] = t;
    }
}

void cwo_set_rng(cwo_env *e, const uint32_t *key, int32_t pos)
{
    memcpy(e->mt, key, sizeof(e->mt));
    e->mti = pos;
}
void cwo_get_rng(const cwo_env *e, uint32_t *key, int32_t *pos)
{
    memcpy(key, e->mt, sizeof(e->mt));
    *pos = e->mti;
}
void cwo_seed_int(cwo_env *e, uint32_t seed) { mt_init_genrand(e, seed); }

/* CPV_COLORS, craftingworld_altobs.py:26-27: pixel k of a cell's 3x3 tile is item k's flag colour
 * (items: 8 objects, then the agent) */
static const uint8_t CPV_COLORS[9][3] = {{45, 82, 160},  {255, 102, 102}, {204, 204, 0},  {211, 211, 211}, {34, 133, 34},
                                         {0, 215, 255},  {153, 52, 255},  {10, 215, 100}, {0, 0, 255}};

static size_t image_bytes(const cwo_config *c)
{
    const size_t s = (size_t)c->size;
    return c->alt_obs ? (3 * s + 3) * (3 * s) * 3 : s * s * 48;
}

/* one cell's tile, altobs.py:527-543 / :625-634: pixel k = (item k present) + (k < 3 and held item k at this
 * cell), times CPV_COLORS[k]; the reference keeps ints (up to 2 x colour), the uint8 view wraps modulo 256 */
static void alt_tile(uint8_t *img, int size, int r, int c, int code, int agent_here, int hold)
{
    const int pw = size * 3;
    for (int k = 0; k < 9; k++) {
        int cnt = (k < 8) ? (code == k + 1) : agent_here;
        if (agent_here && hold != 0 && k == hold - 1) cnt += 1;
        uint8_t *p = img + ((size_t)(r * 3 + k / 3) * pw + (c * 3 + k % 3)) * 3;
        for (int ch = 0; ch < 3; ch++) p[ch] = (uint8_t)(cnt * CPV_COLORS[k][ch]);
    }
}
static void alt_hold_flag(uint8_t *img, int size, int hold)   /* altobs.py:557-559 / :640 */
{
    const int pw = size * 3;
    for (int y = size * 3; y < size * 3 + 3; y++)
        for (int x = 3; x < 6; x++) memset(img + ((size_t)y * pw + x) * 3, hold ? 255 : 0, 3);
}

void cwo_render_alt(int32_t size, const uint8_t *grid, int32_t ar, int32_t ac, int32_t hold, uint8_t *out)
{
    memset(out, 0, (size_t)(3 * size + 3) * (3 * size) * 3);
    for (int r = 0; r < size; r++)
        for (int c = 0; c < size; c++) alt_tile(out, size, r, c, grid[r * size + c], r == ar && c == ac, hold);
    alt_hold_flag(out, size, hold);
}

/* ------------------------------------------------------------------ lifetime */
cwo_env *cwo_new(const cwo_config *cfg)
{
    if (cfg->size < 4 || cfg->size > 255) return NULL; /* >= 12 cells needed by sample_state */
    if (cfg->n_selected < 1 || cfg->n_selected > CWO_MAX_TASKS) return NULL;
    if (cfg->n_task_list < 9 || cfg->n_task_list > CWO_MAX_TASKS) return NULL;
    cwo_env *e = (cwo_env *)calloc(1, sizeof(*e));
    e->cfg = *cfg;
    if (e->cfg.number_of_tasks > e->cfg.n_selected) e->cfg.number_of_tasks = e->cfg.n_selected;
    e->ncell = cfg->size * cfg->size;
    size_t img = image_bytes(cfg);
    e->grid = (uint8_t *)calloc(e->ncell, 1);
    e->init_grid = (uint8_t *)calloc(e->ncell, 1);
    e->goal_grid = (uint8_t *)calloc(e->ncell, 1);
    e->obs = (uint8_t *)calloc(img, 1);
    e->desired_img = (uint8_t *)calloc(img, 1);
    e->init_img = (uint8_t *)calloc(img, 1);
    if (cfg->fixed_init_state > 0) {
        e->pool_grid = (uint8_t *)calloc((size_t)cfg->fixed_init_state * e->ncell, 1);
        e->pool_agent = (int *)calloc((size_t)cfg->fixed_init_state, sizeof(int));
    }
    mt_init_genrand(e, 0);
    return e;
}

void cwo_free(cwo_env *e)
{
    if (!e) return;
    free(e->grid); free(e->init_grid); free(e->goal_grid);
    free(e->obs); free(e->desired_img); free(e->init_img);
    free(e->pool_grid); free(e->pool_agent);
    free(e);
}

/* ------------------------------------------------------------------ render, ray.py:442-520 */
void cwo_render(int32_t size, const uint8_t *grid, int32_t ar, int32_t ac, int32_t hold, uint8_t *out)
{
    const int pw = size * 4; /* pixels per image row */
    /* img = onehot . COLORS_N, then x4 nearest-neighbour along both axes (ray.py:477-479) */
    for (int r = 0; r < size; r++)
        for (int c = 0; c < size; c++) {
            const uint8_t *col = COLORS_N[grid[r * size + c]];
            for (int dy = 0; dy < 4; dy++)
                for (int dx = 0; dx < 4; dx++)
                    memcpy(out + ((size_t)(r * 4 + dy) * pw + (c * 4 + dx)) * 3, col, 3);
        }
    /* agent: centre 2x2 white (ray.py:483) */
    for (int dy = 1; dy < 3; dy++)
        for (int dx = 1; dx < 3; dx++)
            memset(out + ((size_t)(ar * 4 + dy) * pw + (ac * 4 + dx)) * 3, 255, 3);
    /* held object: row 4r+2, cols 4c+1..4c+2 := COLORS_N[hold] (ray.py:484-486) */
    if (hold != 0)
        for (int dx = 1; dx < 3; dx++)
            memcpy(out + ((size_t)(ar * 4 + 2) * pw + (ac * 4 + dx)) * 3, COLORS_N[hold], 3);
}

/* render_edit, ray.py:522-557: repaint one cell of the persistent image in place */
static void render_edit_cell(cwo_env *e, int r, int c)
{
    if (e->cfg.alt_obs) {                          /* altobs.py:625-640 */
        const int here = (r == e->agent_r && c == e->agent_c);
        alt_tile(e->obs, e->cfg.size, r, c, e->grid[r * e->cfg.size + c], here, e->hold);
        if (here) alt_hold_flag(e->obs, e->cfg.size, e->hold);
        return;
    }
    const int size = e->cfg.size, pw = size * 4;
    const uint8_t *col = COLORS_N[e->grid[r * size + c]]; /* np.dot(onehot[:8], COLORS_M), :550 */
    for (int dy = 0; dy < 4; dy++)
        for (int dx = 0; dx < 4; dx++)
            memcpy(e->obs + ((size_t)(r * 4 + dy) * pw + (c * 4 + dx)) * 3, col, 3);
    if (r == e->agent_r && c == e->agent_c) { /* :553 */
        for (int dy = 1; dy < 3; dy++)
            for (int dx = 1; dx < 3; dx++)
                memset(e->obs + ((size_t)(r * 4 + dy) * pw + (c * 4 + dx)) * 3, 255, 3); /* :555 */
        if (e->hold != 0) /* :556-557: 255 - COLORS_H[hold] */
            for (int dx = 1; dx < 3; dx++) {
                uint8_t *p = e->obs + ((size_t)(r * 4 + 2) * pw + (c * 4 + dx)) * 3;
                for (int k = 0; k < 3; k++) p[k] = (uint8_t)(p[k] - COLORS_H[e->hold - 1][k]);
            }
    }
}

/* ------------------------------------------------------------------ sample_state, ray.py:599-628 */
static void sample_state(cwo_env *e, uint8_t *grid, int *agent_cell)
{
    const int n = e->ncell;
    int32_t *perm = (int32_t *)malloc(sizeof(int32_t) * n);
    for (int i = 0; i < n; i++) perm[i] = i;          /* :610 */
    cwo_rng_shuffle(e, perm, n);                       /* :611 */
    /* state = state[perm]: new flat cell k takes old row perm[k]; old rows 0..7 carry objects
     * 0..7 (diag, :605-608), old row 8 carries the agent, rows 9.. are empty */
    for (int k = 0; k < n; k++) {
        int src = perm[k];
        grid[k] = (src < 8) ? (uint8_t)(src + 1) : EMPTY;
        if (src == 8) *agent_cell = k;                 /* :623-626, reshape is row-major (:613) */
    }
    free(perm);
}

void cwo_generate_fixed_states(cwo_env *e) /* ray.py:149-154 */
{
    for (int k = 0; k < e->cfg.fixed_init_state; k++)
        sample_state(e, e->pool_grid + (size_t)k * e->ncell, &e->pool_agent[k]);
}

/* fixed_state_list (ray.py:116-118) as cell indices: out[K][9] = the cells (row*S+col) of objects 0..7 (OBJECTS order) and of the agent */
void cwo_get_fixed_states(const cwo_env *e, uint16_t *out)
{
    for (int k = 0; k < e->cfg.fixed_init_state; k++) {
        const uint8_t *g = e->pool_grid + (size_t)k * e->ncell;
        for (int c = 0; c < e->ncell; c++)
            if (g[c] >= 1 && g[c] <= 8) out[k * 9 + g[c] - 1] = (uint16_t)c;
        out[k * 9 + 8] = (uint16_t)e->pool_agent[k];
    }
}

/* ------------------------------------------------------------------ imagine_obs, ray.py:220-299 */
static int find_nth(const uint8_t *g, int n, int code, int which)
{
    for (int k = 0; k < n; k++)
        if (g[k] == code && which-- == 0) return k;
    return -1;
}
static int count_code(const uint8_t *g, int n, int code)
{
    int cnt = 0;
    for (int k = 0; k < n; k++) cnt += (g[k] == code);
    return cnt;
}
/* k-th cell (row-major) with no object; exclude_cell additionally excluded (agent), or -1 */
static int nth_unoccupied(const uint8_t *g, int n, int exclude_cell, int which)
{
    for (int k = 0; k < n; k++)
        if (g[k] == EMPTY && k != exclude_cell && which-- == 0) return k;
    return -1;
}
static int count_unoccupied(const uint8_t *g, int n, int exclude_cell)
{
    int cnt = 0;
    for (int k = 0; k < n; k++) cnt += (g[k] == EMPTY && k != exclude_cell);
    return cnt;
}

static void imagine_obs(cwo_env *e)
{
    const int n = e->ncell, size = e->cfg.size;
    uint8_t *f = e->goal_grid;
    memcpy(f, e->init_grid, n);                                 /* :225 */
    int agent = e->agent_r * size + e->agent_c;                  /* self.agent_pos */
    const uint32_t d = e->desired;
    if (d & (1u << T_MAKEBREAD)) {                               /* :226-231 first wheat -> bread */
        f[find_nth(f, n, WHEAT, 0)] = BREAD;
    }
    if (d & (1u << T_EATBREAD)) {                                /* :232-237 */
        int which = (int)cwo_rng_randint(e, (uint32_t)count_code(f, n, BREAD));
        f[find_nth(f, n, BREAD, which)] = EMPTY;
    }
    if (d & (1u << T_CHOPTREE)) {                                /* :238-243 first tree -> sticks */
        f[find_nth(f, n, TREE, 0)] = STICKS;
    }
    if (d & (1u << T_MOVESTICKS)) {                              /* :244-257 */
        int which_stick = (int)cwo_rng_randint(e, (uint32_t)count_code(f, n, STICKS));
        /* unoccupied: no object and no agent ([:,:,:9], :252) */
        int which_spot = (int)cwo_rng_randint(e, (uint32_t)count_unoccupied(f, n, agent));
        int from = find_nth(f, n, STICKS, which_stick);
        int to = nth_unoccupied(f, n, agent, which_spot);
        f[from] = EMPTY;
        f[to] = STICKS;
    }
    if (d & (1u << T_BUILDHOUSE)) {                              /* :258-264 */
        int which = (int)cwo_rng_randint(e, (uint32_t)count_code(f, n, STICKS));
        f[find_nth(f, n, STICKS, which)] = HOUSE;
    }
    if (d & (1u << T_CHOPROCK)) {                                /* :265-268 first rock removed */
        f[find_nth(f, n, ROCK, 0)] = EMPTY;
    }
    if (d & (1u << T_GOTOHOUSE)) {                               /* :269-276 agent onto a house */
        int which = (int)cwo_rng_randint(e, (uint32_t)count_code(f, n, HOUSE));
        agent = find_nth(f, n, HOUSE, which);
    }
    if (d & (1u << T_MOVEAXE)) {                                 /* :277-286, agent cell allowed */
        int which_spot = (int)cwo_rng_randint(e, (uint32_t)count_unoccupied(f, n, -1));
        int from = find_nth(f, n, AXE, 0);
        int to = nth_unoccupied(f, n, -1, which_spot);
        f[from] = EMPTY;
        f[to] = AXE;
    }
    if (d & (1u << T_MOVEHAMMER)) {                              /* :287-297 */
        int which_spot = (int)cwo_rng_randint(e, (uint32_t)count_unoccupied(f, n, -1));
        int from = find_nth(f, n, HAMMER, 0);
        int to = nth_unoccupied(f, n, -1, which_spot);
        f[from] = EMPTY;
        f[to] = HAMMER;
    }
    e->goal_agent_r = agent / size;
    e->goal_agent_c = agent % size;
    if (e->cfg.alt_obs) cwo_render_alt(size, f, e->goal_agent_r, e->goal_agent_c, 0, e->desired_img);
    else cwo_render(size, f, e->goal_agent_r, e->goal_agent_c, 0, e->desired_img); /* :299 */
}

/* ------------------------------------------------------------------ reset, ray.py:156-218 */
void cwo_reset(cwo_env *e)
{
    const cwo_config *c = &e->cfg;
    const int n = e->ncell, size = c->size;
    /* task draw, :169-174 */
    int number_of_tasks = c->stacking ? (int)cwo_rng_randint(e, (uint32_t)c->number_of_tasks) + 1 : 1;
    int32_t task_idx[CWO_MAX_TASKS];
    for (int i = 0; i < c->n_selected; i++) task_idx[i] = i;
    cwo_rng_shuffle(e, task_idx, c->n_selected);
    e->desired = 0;
    for (int i = 0; i < number_of_tasks; i++) e->desired |= 1u << c->selected_bits[task_idx[i]];
    e->achieved = 0;                                             /* :176 */
    /* placement, :178-181 */
    int agent_cell = 0;
    if (c->fixed_init_state == 0) {
        sample_state(e, e->grid, &agent_cell);
    } else {                                                     /* :630-644 */
        int k = (int)cwo_rng_randint(e, (uint32_t)c->fixed_init_state);
        memcpy(e->grid, e->pool_grid + (size_t)k * n, n);
        agent_cell = e->pool_agent[k];
    }
    e->agent_r = agent_cell / size;
    e->agent_c = agent_cell % size;
    e->init_agent_r = e->agent_r;
    e->init_agent_c = e->agent_c;
    e->hold = 0;
    memcpy(e->init_grid, e->grid, n);                            /* :183 */
    imagine_obs(e);                                              /* :191 */
    if (c->alt_obs) cwo_render_alt(size, e->grid, e->agent_r, e->agent_c, 0, e->obs);
    else cwo_render(size, e->grid, e->agent_r, e->agent_c, 0, e->obs); /* :192 */
    memcpy(e->init_img, e->obs, image_bytes(c));                 /* :193 */
    if (e->step_num != 0) e->ep_no += 1;                         /* :200-201 */
    e->step_num = 0;                                             /* :203 */
}

/* ------------------------------------------------------------------ eval_task_edit, ray.py:646-703 */
static void eval_task_edit(cwo_env *e, int old_object /* code, or -1 for None */)
{
    const int size = e->cfg.size;
    const int pos = e->agent_r * size + e->agent_c;
    uint32_t a = e->achieved;
    if (old_object == BREAD) a |= 1u << T_EATBREAD;              /* :657-659 */
    else if (old_object == ROCK) a |= 1u << T_CHOPROCK;          /* :660-662 */
    else if (old_object == TREE) a |= 1u << T_CHOPTREE;          /* :663-665 */
    /* :668 assigned, can go 1 -> 0 */
    if (e->grid[pos] == HOUSE) a |= 1u << T_GOTOHOUSE; else a &= ~(1u << T_GOTOHOUSE);
    /* initial contents of this cell; the agent's start cell is "non-empty, not an object"
     * (INIT_OBS_VECTOR keeps channel 8), which takes the same branches as any other object */
    const int init = e->init_grid[pos];
    const int init_is_agent = (e->agent_r == e->init_agent_r && e->agent_c == e->init_agent_c);
    const int init_empty = (init == EMPTY && !init_is_agent);
    if (e->hold == 1) {                                          /* :672-684 holding sticks */
        if (init_empty) a |= 1u << T_MOVESTICKS;
        else if (init == STICKS) a &= ~(1u << T_MOVESTICKS);
        else if (init == TREE && (a & (1u << T_CHOPTREE))) a &= ~(1u << T_MOVESTICKS);
        else a |= 1u << T_MOVESTICKS;
    } else if (e->hold == 2) {                                   /* :685-693 holding axe */
        if (old_object == WHEAT) a |= 1u << T_MAKEBREAD;
        if (init_empty) a |= 1u << T_MOVEAXE;
        else if (init == AXE) a &= ~(1u << T_MOVEAXE);
        else a |= 1u << T_MOVEAXE;
    } else if (e->hold == 3) {                                   /* :694-702 holding hammer */
        if (old_object == STICKS) a |= 1u << T_BUILDHOUSE;
        if (init_empty) a |= 1u << T_MOVEHAMMER;
        else if (init == HAMMER) a &= ~(1u << T_MOVEHAMMER);
        else a |= 1u << T_MOVEHAMMER;
    }
    e->achieved = a;
}

/* compute_reward_equal / _subset, ray.py:747-767 */
static int compute_reward(const cwo_env *e)
{
    const uint32_t full = (1u << e->cfg.n_task_list) - 1u;
    const uint32_t a = e->achieved & full, d = e->desired & full;
    if (!e->cfg.reward_subset)                                   /* short_circuit_check == array_equal */
        return (a == d) ? e->cfg.max_steps : -1;
    /* np.max(desired - achieved) == 0: no (d=1,a=0) position and at least one d == a position */
    int any_missing = (d & ~a) != 0;
    int any_equal = ((~(d ^ a)) & full) != 0;
    return (!any_missing && any_equal) ? e->cfg.max_steps : -1;
}

/* ------------------------------------------------------------------ step, ray.py:301-378 */
int cwo_step(cwo_env *e, int32_t action, int32_t *reward_out, int32_t *done_out)
{
    static const int DR[4] = {-1, 0, 1, 0}, DC[4] = {0, 1, 0, -1}; /* up,right,down,left :130-131 */
    if (action < 0 || action > 5) return -1;
    const int size = e->cfg.size;
    e->step_num += 1;                                            /* :309 */
    int changed = 1;
    int pos = e->agent_r * size + e->agent_c;
    int dirty[2] = {pos, -1};
    if (action == 4) {                                           /* pickup :314-327 */
        int g = e->grid[pos];
        if (!(g == STICKS || g == AXE || g == HAMMER)) changed = 0;
        else if (e->hold != 0) changed = 0;
        else { e->hold = g; e->grid[pos] = EMPTY; }
    } else if (action == 5) {                                    /* drop :329-341 */
        if (e->hold == 0) changed = 0;
        else if (e->grid[pos] != EMPTY) changed = 0;
        else { e->grid[pos] = (uint8_t)e->hold; e->hold = 0; }
    } else {                                                     /* __move_agent :380-440 */
        int old_object = -1;
        int nr = e->agent_r + DR[action], nc = e->agent_c + DC[action];
        nr = nr < 0 ? 0 : (nr > size - 1 ? size - 1 : nr);      /* coord.py:22-25 */
        nc = nc < 0 ? 0 : (nc > size - 1 ? size - 1 : nc);
        if (nr == e->agent_r && nc == e->agent_c) {              /* :395-396 */
            changed = 0;
        } else {
            int npos = nr * size + nc;
            int t = e->grid[npos];
            int cant = (t == ROCK && e->hold != 3) || (t == TREE && e->hold != 2); /* :401-405 */
            if (cant) {
                changed = 0;
            } else {
                dirty[0] = pos; dirty[1] = npos;
                e->agent_r = nr; e->agent_c = nc;                /* :407-410 */
                if (t != EMPTY) {
                    old_object = t;                              /* :411, None when empty (:417-419) */
                    if (t == ROCK || t == BREAD) e->grid[npos] = EMPTY;       /* :423-425 */
                    else if (t == TREE) e->grid[npos] = STICKS;              /* :426-428 */
                    else if (t == STICKS) { if (e->hold == 3) e->grid[npos] = HOUSE; } /* :429-432 */
                    else if (t == WHEAT) { if (e->hold == 2) e->grid[npos] = BREAD; }  /* :433-438 */
                    /* axe, hammer, house: unchanged :420-422 */
                }
            }
        }
        eval_task_edit(e, old_object);                           /* :346, also after failed moves */
    }
    int reward;
    if (changed) {                                               /* :348-361 */
        render_edit_cell(e, dirty[0] / size, dirty[0] % size);
        if (dirty[1] >= 0) render_edit_cell(e, dirty[1] / size, dirty[1] % size);
        reward = compute_reward(e);
    } else {
        reward = -1;                                             /* :363 */
    }
    *reward_out = reward;
    *done_out = (e->step_num >= e->cfg.max_steps || reward == e->cfg.max_steps) ? 1 : 0; /* :367 */
    return 0;
}

void cwo_get_view(const cwo_env *e, cwo_view *v)
{
    v->grid = e->grid; v->init_grid = e->init_grid; v->goal_grid = e->goal_grid;
    v->obs = e->obs; v->desired_img = e->desired_img; v->init_img = e->init_img;
    v->agent_r = e->agent_r; v->agent_c = e->agent_c; v->hold = e->hold;
    v->goal_agent_r = e->goal_agent_r; v->goal_agent_c = e->goal_agent_c;
    v->init_agent_r = e->init_agent_r; v->init_agent_c = e->init_agent_c;
    v->achieved = e->achieved; v->desired = e->desired;
    v->step_num = e->step_num; v->ep_no = e->ep_no;
}

void cwo_set_state(cwo_env *e, const uint8_t *grid, const uint8_t *init_grid, int32_t agent_r,
                   int32_t agent_c, int32_t hold, uint32_t achieved, uint32_t desired, int32_t step_num)
{
    memcpy(e->grid, grid, e->ncell);
    memcpy(e->init_grid, init_grid, e->ncell);
    e->agent_r = agent_r; e->agent_c = agent_c; e->hold = hold;
    e->init_agent_r = -1; e->init_agent_c = -1; /* injected states carry no agent start cell */
    e->achieved = achieved; e->desired = desired; e->step_num = step_num;
    if (e->cfg.alt_obs) cwo_render_alt(e->cfg.size, e->grid, agent_r, agent_c, hold, e->obs);
    else cwo_render(e->cfg.size, e->grid, agent_r, agent_c, hold, e->obs);
}

int64_t cwo_batch_rollout(cwo_env **envs, int32_t n, const int8_t *actions, int32_t T,
                          int32_t nthreads, int32_t *rewards, uint8_t *dones)
{
    int64_t total = 0;
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads) reduction(+ : total) schedule(static)
#endif
    for (int i = 0; i < n; i++) {
        cwo_env *e = envs[i];
        for (int t = 0; t < T; t++) {
            int32_t r, d;
            if (cwo_step(e, actions[(size_t)t * n + i], &r, &d) != 0) continue;
            if (rewards) rewards[(size_t)t * n + i] = r;
            if (dones) dones[(size_t)t * n + i] = (uint8_t)d;
            if (d) cwo_reset(e); /* auto-reset: the next obs is the new episode's first */
            total++;
        }
    }
    (void)nthreads;
    return total;
}

/* Same loop, but every step is followed by render() of the whole frame (ray.py:442-520) instead of relying on the
 * persistent image that render_edit keeps up to date: what "full-frame pixel obs every step" costs on host cores
 * (bench.py's cpu_baseline, beside the reference's own dirty-cell strategy timed by cwo_batch_rollout). */
int64_t cwo_batch_rollout_full(cwo_env **envs, int32_t n, const int8_t *actions, int32_t T, int32_t nthreads)
{
    int64_t total = 0;
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads) reduction(+ : total) schedule(static)
#endif
    for (int i = 0; i < n; i++) {
        cwo_env *e = envs[i];
        for (int t = 0; t < T; t++) {
            int32_t r, d;
            if (cwo_step(e, actions[(size_t)t * n + i], &r, &d) != 0) continue;
            if (d) cwo_reset(e);
            if (e->cfg.alt_obs) cwo_render_alt(e->cfg.size, e->grid, e->agent_r, e->agent_c, e->hold, e->obs);
            else cwo_render(e->cfg.size, e->grid, e->agent_r, e->agent_c, e->hold, e->obs);
            total++;
        }
    }
    (void)nthreads;
    return total;
}
