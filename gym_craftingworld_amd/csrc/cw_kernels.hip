// cw_kernels.hip -- hand-written HIP kernels (gfx950 / CDNA4, wave64) of the CraftingWorld engine.
//
//   cw_step_fused_kernel  step() of ray.py:301-378 for every engine that resets by itself, ONE launch: a wave steps its 8..64 envs on the
//                     sparse slot state (reward, done), a finished env takes over the record of its next episode in its own lane
//                     (LOOK-AHEAD, cw_layout.h), whatever is left -- an env without a record, the frames a reset changes -- by the wave.
//   cw_step_kernel    the same step for engines without auto-reset, one lane per env (in DIRTY pixel mode also render_edit(), ray.py:522-557).
//   cw_refill_kernel  one WAVEFRONT per env: the NEXT reset() of ray.py:156-218, ahead of time and in bulk = task draw, legacy
//                     Fisher-Yates placement on the env's MT19937 stream (state staged in LDS, lane-parallel rejection sampling),
//                     imagine_obs().  cw_reset_kernel: the same for every env at once (explicit reset()).
//   cw_render_pieces_kernel  render() of ray.py:442-520 (and the AltObs raster) for a whole frame ARRAY as a CLOCKED sweep of aligned 4-KiB
//                     pieces: a zero fill plus the few lit bytes of the frames a piece overlaps, at a set rate.  The roofline kernel.
//   cw_rollout_kernel persistent: T steps of every env in one launch (state-only mode).
//   cw_resident_kernel  the single-env loop without a launch per step (doorbell in pinned host memory).
//   cw_export_*       dense grid / one-hot views of the slot state.
//
// All integer; no MFMA (nothing here is a contraction: the reference's tensordot with a one-hot
// operand is a table lookup).  Bounding resource: HBM write bandwidth for cw_render_pieces_kernel,
// issue/latency for the others.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cw_layout.h"
#include "cw_mt.h"

#define CW_WAVE 64
#define CW_BALLOT(p) __builtin_amdgcn_ballot_w64(p)   // the compare's own SGPR pair (__ballot goes through a select + compare)
#define CW_ALT_FRAME_PACE 2    // render_frame_alt: s_sleep(1) after each 1-KiB store of a single frame's zero fill (back to back 3.0 TB/s, with 64-192 idle clocks 5.2-5.4)

// -DCW_TRACE (make trace -> libcraftingworld_trace.so, tools/microbench only): 100 MHz wall-clock stamps of the reset's phases
#ifdef CW_TRACE
__device__ unsigned long long cw_trace_buf[1024 * 8];
#define CW_STAMP(env, k) do { if (lane == 0) { cw_trace_buf[((env) & 1023) * 8 + (k)] = wall_clock64(); \
                                               if ((k) == 0 || (k) == 5) cw_trace_buf[((env) & 1023) * 8 + 6 + ((k) ? 1 : 0)] = clock64(); } } while (0)
extern "C" hipError_t cwk_trace_read(unsigned long long *dst) { return hipMemcpyFromSymbol(dst, HIP_SYMBOL(cw_trace_buf), sizeof(cw_trace_buf)); }
#else
#define CW_STAMP(env, k) do { } while (0)
#endif

enum { EMPTY = 0, STICKS = 1, AXE = 2, HAMMER = 3, ROCK = 4, TREE = 5, BREAD = 6, HOUSE = 7, WHEAT = 8 };
// TASK_LIST bit order, ray.py:40-41
enum { T_MAKEBREAD = 0, T_EATBREAD = 1, T_BUILDHOUSE = 2, T_CHOPTREE = 3, T_CHOPROCK = 4,
       T_GOTOHOUSE = 5, T_MOVEAXE = 6, T_MOVEHAMMER = 7, T_MOVESTICKS = 8 };

typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef u32x3 u32x3_a4 __attribute__((aligned(4)));

__device__ __forceinline__ void unpack_pos(const uint4 &v, uint32_t sp[8])
{
    sp[0] = v.x & 0xFFFFu; sp[1] = v.x >> 16;
    sp[2] = v.y & 0xFFFFu; sp[3] = v.y >> 16;
    sp[4] = v.z & 0xFFFFu; sp[5] = v.z >> 16;
    sp[6] = v.w & 0xFFFFu; sp[7] = v.w >> 16;
}
__device__ __forceinline__ uint4 pack_pos(const uint32_t sp[8])
{
    return make_uint4(sp[0] | (sp[1] << 16), sp[2] | (sp[3] << 16), sp[4] | (sp[5] << 16), sp[6] | (sp[7] << 16));
}
// slot index whose object sits in `cell`, or -1 (at most one object per cell, ray.py:334)
__device__ __forceinline__ int slot_at(const uint32_t sp[8], uint32_t cell)
{
    int idx = -1;
#pragma unroll
    for (int k = 0; k < 8; k++) idx = (sp[k] == cell) ? k : idx;
    return idx;
}
__device__ __forceinline__ uint32_t code_of(uint32_t codes, int idx)
{
    return idx < 0 ? 0u : ((codes >> (4 * idx)) & 15u);
}
__device__ __forceinline__ uint32_t rgb_of_code(uint32_t code)
{
    // COLORS_N (ray.py:28-30) as R | G<<8 | B<<16, index = cell code: a select chain, VALU for a
    // per-lane code and SALU (s_cmp/s_cselect) for a wave-uniform one
    uint32_t c = 0;
    c = code == 1 ? (110u | (69u << 8) | (39u << 16)) : c;
    c = code == 2 ? (255u | (105u << 8) | (180u << 16)) : c;
    c = code == 3 ? (100u | (100u << 8) | (200u << 16)) : c;
    c = code == 4 ? (100u | (100u << 8) | (100u << 16)) : c;
    c = code == 5 ? (0u | (128u << 8) | (0u << 16)) : c;
    c = code == 6 ? (205u | (133u << 8) | (63u << 16)) : c;
    c = code == 7 ? (197u | (91u << 8) | (97u << 16)) : c;
    c = code == 8 ? (240u | (230u << 8) | (140u << 16)) : c;
    return c;
}

// One cell's 4 pixels of one pixel row = 12 bytes R G B R | G B R G | B R G B.
__device__ __forceinline__ u32x3 cell_row_dwords(uint32_t rgb)
{
    u32x3 d;
    d.x = rgb | (rgb << 24);
    d.y = (rgb >> 8) | (rgb << 16);
    d.z = (rgb >> 16) | (rgb << 8);
    return d;
}
// agent overlay on pixels 1,2 of the cell row (bytes 3..8): ray.py:483-486 / :555-557
__device__ __forceinline__ u32x3 overlay_dwords(u32x3 d, uint32_t o)
{
    d.x = (d.x & 0x00FFFFFFu) | (o << 24);
    d.y = (o >> 8) | (o << 16);
    d.z = (d.z & 0xFFFFFF00u) | (o >> 16);
    return d;
}

// render_edit (ray.py:522-557): repaint one cell of the persistent frame, one lane does 4 x 12 B
__device__ __forceinline__ void paint_cell(uint8_t *frame, int S, uint32_t cell, uint32_t code,
                                           bool agent_here, uint32_t hold, uint32_t div_magic, bool mark_rows_only = false)
{
    const uint32_t r = __umulhi(cell, div_magic);
    const uint32_t c = cell - r * S;
    const uint32_t rgb = rgb_of_code(code);
    const u32x3 base = cell_row_dwords(rgb);
    const uint32_t row_bytes = 12u * S;
    uint8_t *p = frame + (size_t)(4u * r) * row_bytes + 12u * c;
#pragma unroll
    for (int dy = 0; dy < 4; dy++) {
        u32x3 d = base;
        if (agent_here && (dy == 1 || dy == 2)) {
            uint32_t o = (dy == 2 && hold != 0) ? rgb_of_code(hold) : 0x00FFFFFFu;
            d = overlay_dwords(d, o);
        }
        if (dy == 1 || dy == 2 || !mark_rows_only) *(u32x3_a4 *)(p + dy * row_bytes) = d;      // (rows 0, 3 never carry the mark)
    }
}

// ------------------------------------------------------------------------------------ AltObs rasteriser
// CraftingWorldEnvAltObs (craftingworld_altobs.py:489-642): 3x3 px per cell; pixel k of the tile carries
// CPV_COLORS[k] (altobs.py:26-27) times the number of items k in the cell, items = 8 objects + agent, the
// held object counted on its own object pixel at the agent cell (so sticks held over sticks give 2 x colour:
// the reference's int image, here modulo 256).  Frame = [3S+3][3S][3] bytes; the last 3 pixel rows are a
// strip whose pixels 1..1 (bytes 9..17 of each row) are white while something is held (altobs.py:557-559).
// 27*S*(S+1) bytes is even but not a multiple of 4 and rows are 9S bytes, so tiles are written bytewise.
__device__ __forceinline__ uint32_t cpv_color(int k)       // R | G<<8 | B<<16
{
    return k == 0 ? (45u | (82u << 8) | (160u << 16)) : k == 1 ? (255u | (102u << 8) | (102u << 16))
         : k == 2 ? (204u | (204u << 8)) : k == 3 ? (211u | (211u << 8) | (211u << 16))
         : k == 4 ? (34u | (133u << 8) | (34u << 16)) : k == 5 ? ((215u << 8) | (255u << 16))
         : k == 6 ? (153u | (52u << 8) | (255u << 16)) : k == 7 ? (10u | (215u << 8) | (100u << 16)) : (255u << 16);
}
__device__ __forceinline__ void alt_paint_tile(uint8_t *frame, int S, uint32_t r, uint32_t c, uint32_t code,
                                               bool agent_here, uint32_t hold)
{
    const uint32_t row_bytes = 9u * S;
    uint8_t *p = frame + (size_t)(3u * r) * row_bytes + 9u * c;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        uint32_t cnt = (k < 8) ? (code == (uint32_t)(k + 1) ? 1u : 0u) : (agent_here ? 1u : 0u);
        if (k < 3) cnt += (agent_here && hold == (uint32_t)(k + 1)) ? 1u : 0u;
        const uint32_t col = cpv_color(k);
        uint8_t *q = p + (k / 3) * row_bytes + (k % 3) * 3;
        q[0] = (uint8_t)(cnt * (col & 0xFFu));
        q[1] = (uint8_t)(cnt * ((col >> 8) & 0xFFu));
        q[2] = (uint8_t)(cnt * ((col >> 16) & 0xFFu));
    }
}
// the "holding" strip: 3 rows x 9S bytes after the grid; `first`/`step` let one lane or a whole wave write it
__device__ __forceinline__ void alt_paint_strip(uint8_t *frame, int S, uint32_t hold, uint32_t first, uint32_t step,
                                                bool whole)
{
    const uint32_t row_bytes = 9u * S;
    uint8_t *p = frame + (size_t)(3u * S) * row_bytes;
    if (whole) {
        for (uint32_t b = first; b < 3u * row_bytes; b += step) {
            const uint32_t x = b % row_bytes;
            p[b] = (hold != 0 && x >= 9u && x < 18u) ? 255 : 0;
        }
    } else {      // only the 27 flag bytes (the rest of the strip never changes)
        for (uint32_t j = first; j < 27u; j += step) p[(j / 9u) * row_bytes + 9u + (j % 9u)] = hold ? 255 : 0;
    }
}
// A whole AltObs frame by one wavefront.  The frame is almost all zeros: each of the <= 10 items lights ONE pixel (3 bytes)
// of its cell's tile, and the "holding" flag is 9 more pixels.  So a frame is a zero FILL -- 16-byte chunks aligned in
// memory (frames are 27*S*(S+1) bytes: even, not a multiple of 4, so every frame starts at another alignment; the <= 15
// bytes before the first and after the last aligned chunk go out as single bytes, one lane each) -- followed by the <= 19
// three-byte pixels, lane p = pixel p, three byte stores.  The pixel stores come after the fill stores of the same wave to
// the same addresses: one wave's stores to one address are performed in program order.  Two items on one pixel give
// 2 x colour in the reference's int image (sticks held over sticks, altobs.py:527-543), here modulo 256: the only pixels
// that can coincide are the held item's and an object's, and they are merged before they are stored.
// (The tile-by-tile form, 27 byte stores per cell, reached 4.3 TB/s on L2 write combining.)
__device__ __forceinline__ void render_frame_alt(uint8_t *__restrict__ dst0, uint8_t *__restrict__ dst1, int S, int ncell,
                                                 uint32_t div_magic, const uint32_t sp[8], uint32_t codes,
                                                 uint32_t agent_cell, uint32_t hold, int lane, int pace)
{
    const uint32_t row_bytes = 9u * S, FB = 27u * S * (uint32_t)(S + 1);
    // ---- the frame's lit pixels, lane p = pixel p: byte offset in the frame (none: 0xFFFFFFFF) and R | G<<8 | B<<16
    uint32_t pos = 0xFFFFFFFFu, item = 0;
#pragma unroll
    for (int q = 0; q < 8; q++) {                             // lanes 0..7: object slots (pixel = code - 1 of the slot's cell)
        pos = (lane == q) ? sp[q] : pos;
        item = (lane == q) ? ((codes >> (4 * q)) & 15u) : item;
    }
    if (lane == 8) { pos = agent_cell; item = 9u; }          // the agent: pixel 8 (altobs.py:536)
    if (lane == 9) { pos = agent_cell; item = hold; }        // the held item, on its own object pixel at the agent's cell
    uint32_t p_off = 0xFFFFFFFFu, p_val = 0;
    if (lane < 10 && item != 0 && pos < (uint32_t)ncell) {
        const uint32_t r = __umulhi(pos, div_magic), c = pos - r * S, k = item - 1u;
        const uint32_t k3 = (k >= 6u) ? 2u : (k >= 3u) ? 1u : 0u;
        p_off = (3u * r + k3) * row_bytes + 9u * c + 3u * (k - 3u * k3);
        p_val = cpv_color((int)k);
    } else if (lane >= 10 && lane < 19 && hold != 0) {      // the strip's flag: pixels 3..5 of its three rows (altobs.py:557-559)
        const uint32_t j = (uint32_t)lane - 10u, jr = (j >= 6u) ? 2u : (j >= 3u) ? 1u : 0u;
        p_off = (3u * S + jr) * row_bytes + 9u + 3u * (j - 3u * jr);
        p_val = 0x00FFFFFFu;
    }
    // the held item's pixel on top of an object's: one store of the sum, byte-wise modulo 256
    const uint32_t held_off = __builtin_amdgcn_readlane(p_off, 9);
    const bool twice = lane < 8 && p_off != 0xFFFFFFFFu && p_off == held_off;
    if (twice) p_val = ((2u * (p_val & 0xFFu)) & 0xFFu) | ((2u * (p_val & 0xFF00u)) & 0xFF00u) | ((2u * (p_val & 0xFF0000u)) & 0xFF0000u);
    if (CW_BALLOT(twice) && lane == 9) p_off = 0xFFFFFFFFu;
    const bool p_any = p_off != 0xFFFFFFFFu;
    for (int which = 0; which < 2; which++) {
        uint8_t *const dst = which ? dst1 : dst0;
        if (!dst) break;
        const uint32_t head = (uint32_t)(-(intptr_t)dst) & 15u;               // bytes before the first aligned chunk
        const uint32_t n_body = (FB - head) >> 4;                              // aligned 16-B chunks inside the frame
        const uint32_t tail0 = head + 16u * n_body;                            // first byte after them
        if ((uint32_t)lane < head) dst[lane] = 0;
        if (tail0 + (uint32_t)lane < FB) dst[tail0 + lane] = 0;
        for (uint32_t j = (uint32_t)lane; j < n_body; j += CW_WAVE) {
            *(uint4 *)(dst + head + 16u * j) = make_uint4(0, 0, 0, 0);
            // PACING (see render_groups): 1 KiB stores back to back run at 3.0 TB/s, with 64-192 idle clocks after each at 5.2-5.4
            for (int z = 0; z < pace; z++) __builtin_amdgcn_s_sleep(1);
        }
        if (p_any) {
            dst[p_off] = (uint8_t)p_val;
            dst[p_off + 1] = (uint8_t)(p_val >> 8);
            dst[p_off + 2] = (uint8_t)(p_val >> 16);
        }
    }
}

// ------------------------------------------------------------------------------------ single frames
// One wavefront paints one frame, 64 cells per iteration: lane = one cell (row-major), whose colour
// is an 8-compare chain against the wave-uniform slot positions (SGPRs), computed ONCE and stored
// to the cell's 4 pixel rows as 4 x 12 B (global_store_dwordx3).  This is the painter of single frames
// -- the INIT_OBS / desired_goal frames of finished envs and their terminal frames (cw_step_fused_kernel), all three frames of an env
// reset inside the dirty-cell step, cw_render into an array the sweep's 16-byte stores cannot take (cw_render_frames_kernel); every
// whole ARRAY is swept (render_pieces below).
// Plain stores: nontemporal ones measured 25 % slower in this shape (tools/microbench).
__device__ __forceinline__ void render_frame(uint8_t *__restrict__ dst0, uint8_t *__restrict__ dst1,
                                             int S, int ncell, uint32_t div_magic, const uint32_t sp[8],
                                             const uint32_t rgb[8], uint32_t agent_cell, uint32_t hold_rgb,
                                             int lane)
{
    const uint32_t row_bytes = 12u * S;
    for (uint32_t cell = lane; cell < (uint32_t)ncell; cell += CW_WAVE) {
        const uint32_t r = __umulhi(cell, div_magic);
        const uint32_t c = cell - r * S;
        uint32_t col = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) col = (cell == sp[k]) ? rgb[k] : col;
        const u32x3 d = cell_row_dwords(col);
        const bool ag = (cell == agent_cell);
        const u32x3 d1 = ag ? overlay_dwords(d, 0x00FFFFFFu) : d;     // ray.py:483
        const u32x3 d2 = ag ? overlay_dwords(d, hold_rgb) : d;        // ray.py:484-486
        const size_t off = (size_t)(4u * r) * row_bytes + 12u * c;
        uint8_t *q = dst0 + off;
        *(u32x3_a4 *)(q) = d;
        *(u32x3_a4 *)(q + row_bytes) = d1;
        *(u32x3_a4 *)(q + 2 * row_bytes) = d2;
        *(u32x3_a4 *)(q + 3 * row_bytes) = d;
        if (dst1) {
            uint8_t *q1 = dst1 + off;
            *(u32x3_a4 *)(q1) = d;
            *(u32x3_a4 *)(q1 + row_bytes) = d1;
            *(u32x3_a4 *)(q1 + 2 * row_bytes) = d2;
            *(u32x3_a4 *)(q1 + 3 * row_bytes) = d;
        }
    }
}
// one frame of either raster from wave-uniform values: slot positions / codes, agent cell, hold
__device__ __forceinline__ void paint_state_frame(const CwParams &P, uint8_t *dst, const uint32_t sp[8], uint32_t codes, uint32_t agent_cell,
                                                  uint32_t hold, int lane, uint8_t *dst1 = nullptr)
{
    if (P.raster == 1) {
        render_frame_alt(dst, dst1, P.size, P.ncell, P.div_magic, sp, codes, agent_cell, hold, lane, CW_ALT_FRAME_PACE);
    } else {
        uint32_t rgb[8];
#pragma unroll
        for (int k = 0; k < 8; k++) rgb[k] = rgb_of_code((codes >> (4 * k)) & 15u);
        render_frame(dst, dst1, P.size, P.ncell, P.div_magic, sp, rgb, agent_cell, hold ? rgb_of_code(hold) : 0x00FFFFFFu, lane);
    }
}
// which of an env's three states a frame shows: its current one (observation), the one at reset (INIT_OBS), imagine_obs' final one (desired_goal)
enum { CW_SRC_CURRENT = 0, CW_SRC_INIT = 1, CW_SRC_GOAL = 2 };
// ... as wave-uniform values (every lane loads the same record)
__device__ __forceinline__ void load_state_uniform(const CwParams &P, int env, int src, uint32_t sp[8], uint32_t &codes, uint32_t &agent_cell, uint32_t &hold)
{
    uint4 v_p;
    uint32_t v_c, v_a, v_h = 0;
    if (src == CW_SRC_CURRENT) {
        const uint4 h = P.hdr[env];
        v_p = P.pos[env];
        v_c = h.w;
        v_a = (h.x & 0xFFu) * P.size + ((h.x >> 8) & 0xFFu);
        v_h = (h.x >> 16) & 0xFFu;
    } else if (src == CW_SRC_INIT) {
        v_p = P.init_pos[env];
        v_c = CW_CODES_INITIAL;
        v_a = P.init_agent[env];
    } else {
        v_p = P.goal_pos[env];
        v_c = P.goal_codes[env];
        v_a = P.goal_agent[env];
    }
    unpack_pos(make_uint4(__builtin_amdgcn_readfirstlane(v_p.x), __builtin_amdgcn_readfirstlane(v_p.y), __builtin_amdgcn_readfirstlane(v_p.z),
                          __builtin_amdgcn_readfirstlane(v_p.w)), sp);
    codes = __builtin_amdgcn_readfirstlane(v_c);
    agent_cell = __builtin_amdgcn_readfirstlane(v_a);
    hold = __builtin_amdgcn_readfirstlane(v_h);
}

// ------------------------------------------------------------------------------------ step
// One env's step() on registers (ray.py:301-378).  Shared by cw_step_kernel (one launch per step)
// and cw_rollout_kernel (T steps in one persistent launch).
struct CwStepOut {
    int reward;
    bool done, success, invalid, changed;
    uint32_t dirty0, dirty1;       // cells to repaint (render_edit): dirty1 = 0xFFFFFFFF if only one
    bool mark0, mark1;             // only the agent's mark came or went in that cell (a move that left the cell's object as it was): pixel rows 1, 2 suffice
    uint32_t step_num, achieved, desired;
    uint32_t n_success;            // steps of this episode that returned MAX_STEPS so far, this one included (header flags bits 2-15)
};

// The episode's RETURN so far as the reference's loop sums it (ray.py:361-367: a step returns MAX_STEPS when it leaves the goal satisfied, else -1):
// n_success x MAX_STEPS - (step_num - n_success).  An engine that resets by itself ends the episode at its first done: MAX_STEPS - (step_num - 1)
// after a success, -step_num after a time-out.  One WITHOUT auto-reset keeps stepping a finished env, as the reference does (ray.py:367), and a goal
// that stays satisfied pays again on every step that changes the state: the header counts those steps (14 bits, saturating).
__device__ __forceinline__ int32_t episode_return_of(const CwParams &P, const CwStepOut &o)
{
    return (int32_t)o.n_success * (P.max_steps + 1) - (int32_t)o.step_num;
}

template <typename InitPosFn>
__device__ __forceinline__ CwStepOut step_env(const CwParams &P, uint4 &h, uint32_t sp[8], int a, InitPosFn load_init_pos)
{
    CwStepOut o;
    const int S = P.size;
    int ar = h.x & 0xFF, ac = (h.x >> 8) & 0xFF;
    uint32_t hold = (h.x >> 16) & 0xFF;
    uint32_t achieved = h.y & 0xFFFFu;
    const uint32_t desired = h.y >> 16;
    // ray.py:309; 16-bit field, saturating: a finished env stepped on without auto-reset stays done (max_steps <= 65535)
    const uint32_t step_num = min((h.z & 0xFFFFu) + 1u, 0xFFFFu);
    const uint32_t flags = (h.z >> 16) & ~CW_FLAG_RESET;
    uint32_t codes = h.w;

    o.invalid = (unsigned)a > 5u;
    bool changed = false;
    const uint32_t cell = ar * S + ac;
    const int idx_here = slot_at(sp, cell);
    const uint32_t code_here = code_of(codes, idx_here);
    o.dirty0 = cell;
    o.dirty1 = 0xFFFFFFFFu;
    o.mark0 = o.mark1 = false;

    if (a == 4) {                                             // pickup, ray.py:314-327
        if (code_here >= STICKS && code_here <= HAMMER && hold == 0) {
            hold = code_here;
#pragma unroll
            for (int k = 0; k < 8; k++) sp[k] = (k == idx_here) ? CW_POS_HELD : sp[k];
            changed = true;
        }
    } else if (a == 5) {                                      // drop, ray.py:329-341
        if (hold != 0 && idx_here < 0) {
#pragma unroll
            for (int k = 0; k < 8; k++) sp[k] = (sp[k] == CW_POS_HELD) ? cell : sp[k];
            hold = 0;
            changed = true;
        }
    } else if (!o.invalid) {                                  // __move_agent, ray.py:380-440
        const int dr = (a == 0) ? -1 : (a == 2) ? 1 : 0;      // ACTIONS = up,right,down,left, :130-131
        const int dc = (a == 1) ? 1 : (a == 3) ? -1 : 0;
        const int nr = min(max(ar + dr, 0), S - 1);           // Coord.__add__, coord.py:22-25
        const int nc = min(max(ac + dc, 0), S - 1);
        uint32_t old_obj = 0;                                 // None
        uint32_t cur_code = code_here;
        if (nr != ar || nc != ac) {                           // :395-396
            const uint32_t ncell = nr * S + nc;
            const int idx_t = slot_at(sp, ncell);
            const uint32_t t = code_of(codes, idx_t);
            const bool cant = (t == ROCK && hold != 3) || (t == TREE && hold != 2);  // :401-405
            if (!cant) {
                changed = true;
                o.dirty1 = ncell;
                o.mark0 = true;                               // (what lies in the cell it leaves stays)
                ar = nr; ac = nc;
                old_obj = t;                                  // :411 (None when empty, :417-419)
                uint32_t nw = t;
                if (t == ROCK || t == BREAD) nw = EMPTY;      // :423-425
                else if (t == TREE) nw = STICKS;              // :426-428
                else if (t == STICKS && hold == 3) nw = HOUSE; // :429-432
                else if (t == WHEAT && hold == 2) nw = BREAD;  // :433-438
                if (nw != t) {
                    codes = (codes & ~(15u << (4 * idx_t))) | (nw << (4 * idx_t));
                    if (nw == EMPTY) {
#pragma unroll
                        for (int k = 0; k < 8; k++) sp[k] = (k == idx_t) ? CW_POS_GONE : sp[k];
                    }
                }
                cur_code = nw;
                o.mark1 = nw == t;
            }
        }
        // eval_task_edit, ray.py:646-703 -- runs after every move action, failed ones included
        const uint32_t pcell = ar * S + ac;
        if (old_obj == BREAD) achieved |= 1u << T_EATBREAD;            // :657-659
        else if (old_obj == ROCK) achieved |= 1u << T_CHOPROCK;        // :660-662
        else if (old_obj == TREE) achieved |= 1u << T_CHOPTREE;        // :663-665
        achieved = (cur_code == HOUSE) ? (achieved | (1u << T_GOTOHOUSE))
                                       : (achieved & ~(1u << T_GOTOHOUSE));  // :668
        if (hold != 0) {
            const uint4 ip = load_init_pos();
            const uint32_t ip_sticks = ip.x & 0xFFFFu, ip_axe = ip.x >> 16;
            const uint32_t ip_hammer = ip.y & 0xFFFFu, ip_tree = ip.z & 0xFFFFu;
            if (hold == 1) {                                           // :672-684
                const bool home = (pcell == ip_sticks) ||
                                  (pcell == ip_tree && (achieved & (1u << T_CHOPTREE)));
                achieved = home ? (achieved & ~(1u << T_MOVESTICKS)) : (achieved | (1u << T_MOVESTICKS));
            } else if (hold == 2) {                                    // :685-693
                if (old_obj == WHEAT) achieved |= 1u << T_MAKEBREAD;
                achieved = (pcell == ip_axe) ? (achieved & ~(1u << T_MOVEAXE)) : (achieved | (1u << T_MOVEAXE));
            } else {                                                   // :694-702
                if (old_obj == STICKS) achieved |= 1u << T_BUILDHOUSE;
                achieved = (pcell == ip_hammer) ? (achieved & ~(1u << T_MOVEHAMMER)) : (achieved | (1u << T_MOVEHAMMER));
            }
        }
    }

    // reward, ray.py:348-363 + 747-767
    int reward = -1;
    if (changed) {
        const uint32_t am = achieved & P.task_mask, dm = desired & P.task_mask;
        bool hit;
        if (flags & CW_FLAG_SUBSET)  // np.max(desired - achieved) == 0
            hit = ((dm & ~am) == 0) && (((~(dm ^ am)) & P.task_mask) != 0);
        else                         // short_circuit_check == array_equal
            hit = (am == dm);
        reward = hit ? P.max_steps : -1;
    }
    o.reward = reward;
    o.changed = changed;
    o.success = (reward == P.max_steps);
    o.done = (step_num >= (uint32_t)P.max_steps) || o.success;             // :367
    o.step_num = step_num;
    o.achieved = achieved;
    o.desired = desired;
    o.n_success = min((flags >> 2) + (o.success ? 1u : 0u), 0x3FFFu);

    h.x = (uint32_t)ar | ((uint32_t)ac << 8) | (hold << 16) | (h.x & 0xFF000000u);
    h.y = achieved | (desired << 16);
    h.z = step_num | (((flags & 3u) | (o.n_success << 2)) << 16);
    h.w = codes;
    return o;
}

// render_edit (ray.py:522-557 / altobs.py:625-640): repaint the <= 2 cells a step changed in the env's persistent frame, by the env's lane
__device__ __forceinline__ void paint_changed_cells(const CwParams &P, int env, const uint4 &h, const uint32_t sp[8], const CwStepOut &o)
{
    uint8_t *frame = P.obs + (size_t)env * P.frame_bytes;
    const uint32_t hold = (h.x >> 16) & 0xFFu;
    const uint32_t acell = (h.x & 0xFFu) * P.size + ((h.x >> 8) & 0xFFu);
    if (P.raster == 1) {
        const uint32_t r0 = __umulhi(o.dirty0, P.div_magic);
        alt_paint_tile(frame, P.size, r0, o.dirty0 - r0 * P.size, code_of(h.w, slot_at(sp, o.dirty0)), o.dirty0 == acell, hold);
        if (o.dirty1 != 0xFFFFFFFFu) {
            const uint32_t r1 = __umulhi(o.dirty1, P.div_magic);
            alt_paint_tile(frame, P.size, r1, o.dirty1 - r1 * P.size, code_of(h.w, slot_at(sp, o.dirty1)), o.dirty1 == acell, hold);
        }
        alt_paint_strip(frame, P.size, hold, 0, 1, false);
    } else {
        paint_cell(frame, P.size, o.dirty0, code_of(h.w, slot_at(sp, o.dirty0)), o.dirty0 == acell, hold, P.div_magic, o.mark0);
        if (o.dirty1 != 0xFFFFFFFFu)
            paint_cell(frame, P.size, o.dirty1, code_of(h.w, slot_at(sp, o.dirty1)), o.dirty1 == acell, hold, P.div_magic, o.mark1);
    }
}

// LOOK-AHEAD (cw_layout.h): the finished env `env` takes over the record of its next episode, if the refill kernel has left one -- the whole
// of reset() (ray.py:156-218) as three 16-byte loads and the stores of the episode records, by the lane that stepped the env.  h / sp become
// the new episode's header and slots (reset_header's values).  -> false: no record (the env finished twice between two refills, or the
// engine keeps none); the caller hands the env to the slow path, which resets it from the same position of its stream.
// ctl: the env's nx_ctl word (head slot | CW_CTL_QUEUED), loaded by the caller with the env's state.  Taking a record marks the env QUEUED: the next refill
// kernel, which scans these words, tops its ring up (there is no list to append to, and so no ticket to wait for).
struct CwGoalState { uint4 pos; uint32_t codes, agent; };      // imagine_obs' final state of the episode just taken over (the painters of its desired_goal frame)
__device__ __forceinline__ bool pop_next_episode(const CwParams &P, int env, uint32_t ctl, uint4 &h, uint32_t sp[8], bool count_episode, CwGoalState *goal = nullptr)
{
    const uint32_t slot = ctl & 0xFFu;
    const size_t at = (size_t)slot * P.n_envs + env;
    // (everything the take-over reads is asked for at once -- the record's three parts and the episode counter: ONE memory round trip; a load issued behind the
    // valid bit's test would be a second one on the critical path of every wave with a finished env)
    const uint4 m = P.nx_misc[at];
    const uint4 ipos = P.nx_init_pos[at];
    const uint4 gpos = P.nx_goal_pos[at];
    const int32_t ep_no = count_episode ? P.ep_no[env] : 0;
    if (!(m.z >> 31)) return false;
    P.init_pos[env] = ipos;
    P.goal_pos[env] = gpos;
    if (goal) { goal->pos = gpos; goal->codes = m.y; goal->agent = m.x >> 16; }
    P.goal_codes[env] = m.y;
    P.init_agent[env] = (uint16_t)(m.x & 0xFFFFu);
    P.goal_agent[env] = (uint16_t)(m.x >> 16);
    if (count_episode) P.ep_no[env] = ep_no + 1;                  // ray.py:200-201
    ((uint32_t *)(P.nx_misc + at))[2] = 0;                        // taken: the slot is free, the next one (if any waits) is the head
    P.nx_ctl[env] = (slot + 1u == CW_LA_DEPTH ? 0u : slot + 1u) | CW_CTL_QUEUED;      // (on the list: the caller has put it there, or it was there)
    const uint32_t ia = m.x & 0xFFFFu;
    const uint32_t ar = __umulhi(ia, P.div_magic), ac = ia - ar * P.size;
    h.x = ar | (ac << 8) | (h.x & 0xFF000000u);                   // (the menu id stays)
    h.y = (m.z & 0xFFFFu) << 16;
    h.z = (CW_FLAG_RESET | (((m.z >> 16) & 1u) ? CW_FLAG_SUBSET : 0u)) << 16;
    h.w = CW_CODES_INITIAL;
    unpack_pos(ipos, sp);
    return true;
}

// An env reset the SLOW way holds no record for its next episode: it must be QUEUED for the next refill -- otherwise an engine whose host never asks for a
// whole-batch refill again (a captured graph replayed after a re-seed dropped every record: cw_refill_kernel with all_envs = 0 baked in) would reset the slow
// way for the rest of its life.  Its QUEUED bit and the count of slow resets (what the host's refill period follows).  One lane.
__device__ __forceinline__ void note_slow_reset(const CwParams &P, int env, uint32_t ctl)
{
    atomicAdd(&P.counters[5], 1ull);                  // (private word: resets taken the slow way)
    if (!(ctl & CW_CTL_QUEUED)) P.nx_ctl[env] = ctl | CW_CTL_QUEUED;
}

// step() for engines WITHOUT auto-reset (the single-env loop's launch path, fixture replays): one lane per env, finished envs keep
// stepping until cw_reset (ray.py:367); in DIRTY pixel mode also render_edit() of the <= 2 changed cells.  Engines that reset by
// themselves run cw_step_fused_kernel.
__global__ __launch_bounds__(256) void cw_step_kernel(CwParams P, const void *actions, int act_dtype, int paint_dirty)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) {
        atomicAdd(&P.counters[0], (unsigned long long)P.n_envs);
    }
    const bool live = i < P.n_envs;
    bool done = false, success = false, invalid = false;
    if (live) {
        int a;
        if (act_dtype == 0) a = ((const int32_t *)actions)[i];
        else if (act_dtype == 1) a = (int)((const long long *)actions)[i];
        else a = ((const uint8_t *)actions)[i];

        uint4 h = P.hdr[i];
        uint32_t sp[8];
        unpack_pos(P.pos[i], sp);
        const CwStepOut o = step_env(P, h, sp, a, [&]() { return P.init_pos[i]; });
        done = o.done; success = o.success; invalid = o.invalid;

        P.hdr[i] = h;
        P.pos[i] = pack_pos(sp);
        P.reward[i] = o.reward;
        P.done[i] = done ? 1 : 0;
        P.achieved_out[i] = (uint16_t)o.achieved;
        P.desired_out[i] = (uint16_t)o.desired;
        if (done) { P.episode_length[i] = (int32_t)o.step_num; P.episode_return[i] = episode_return_of(P, o); }
        if (paint_dirty && o.changed) paint_changed_cells(P, i, h, sp, o);   // render_edit, :358
    }
    const unsigned long long m_done = CW_BALLOT(done), m_succ = CW_BALLOT(success), m_inv = CW_BALLOT(invalid);
    if ((m_done | m_inv) && (threadIdx.x & (CW_WAVE - 1)) == 0) {
        if (m_done) atomicAdd(&P.counters[1], (unsigned long long)__popcll(m_done));
        if (m_succ) atomicAdd(&P.counters[2], (unsigned long long)__popcll(m_succ));
        if (m_inv) atomicAdd(&P.counters[3], (unsigned long long)__popcll(m_inv));
    }
}

// ------------------------------------------------------------------------------------ reset
// One WAVEFRONT per finished env.  reset() is an inherently serial walk down the env's MT19937
// stream (a rejection-sampled Fisher-Yates of S*S cells, then imagine_obs), so a lane-per-env
// kernel leaves every global-memory latency of that stream on the critical path -- measured 127 us
// per reset alone and 350 us beside the render kernel's HBM traffic.  Here the wave stages the
// env's 624-word state in LDS with coalesced loads, regenerates it 64 words per parallel step
// (CwMtWave::gen), and runs the serial consumer on WAVE-UNIFORM values: each raw draw is a
// v_readlane of the chunk register, the bookkeeping compiles to scalar (SALU) code, and the 9
// token positions live in lanes 0..8 of one VGPR.  No global access is left inside
// the chain, so the reset costs the same whether or not the chip is busy writing frames.

// Fisher-Yates of arange(ncell) (RandomState.shuffle, ray.py:610-612) tracking only the 9 tokens
// that matter: values 0..7 = objects, 8 = agent (diag rows, ray.py:605-608).  All tokens start
// at positions 0..8; position i > 8 holds a non-token until its own swap and is final after it,
// so a 9-nibble map "which token sits at low position q" is all the state the loop needs.
// new_cell[k] = old_row[perm[k]] means token v ends in the cell whose perm entry is v.
// Returns a VGPR whose lane v (0..8) holds token v's final cell.
__device__ __forceinline__ uint32_t shuffle_tokens(CwMtWave &mt, int n)
{
    const uint32_t lane = mt.lane;
    uint32_t hit = 0;                               // bit q: low position q has been swapped out (holds a non-token)
    uint32_t v_tok = 0;
    int i = n - 1;

    // ---- lane-parallel phase: up to 64 raw draws per round, as long as every i stays > 8.
    // Draw l of a chunk is accepted iff (d_l & mask(i_l)) <= i_l with i_l = i - #accepted before l:
    // a prefix dependency, resolved by iterating acc -> F(acc) to its fixed point.  F is strictly
    // lower-triangular (lane l depends on lanes < l only), so the fixed point is unique, reached from any
    // start, and lane l is exact after l+1 passes at the latest; starting from "accepted against the
    // round's first i" a draw's fate only flips if its value sits within a few counts of i_l, so one or
    // two passes settle all 64.  The sequential semantics are reproduced exactly; only the rare accepted
    // draws that hit a low position (v <= 8) are then replayed in lane order on the scalar unit.  In this
    // phase position i > 8 always holds a non-token, so low position q holds either its original token q
    // or (after its first hit) a non-token: one bit per position is the whole state.  A round takes at
    // most i-8 draws, so even if all of them are accepted no i_l drops to 8; the rest of the chunk stays
    // for the next round.
    while (i >= 10) {
        if (mt.used == 64) mt.gen();
        const int take = min(64 - mt.used, i - 8);
        const uint32_t d = mt.v_out;
        // lanes [used, used+take) hold this round's draws (take >= 1)
        const unsigned long long valid = (~0ull >> (64 - take)) << mt.used;
        unsigned long long acc = CW_BALLOT((d & (0xFFFFFFFFu >> __builtin_clz((uint32_t)i))) <= (uint32_t)i) & valid;
        uint32_t i_l = 0, v_l = 0;
        for (;;) {                                   // <= 64 passes (lane l is exact after l+1), typically 2
            const uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(acc >> 32),
                                                              __builtin_amdgcn_mbcnt_lo((uint32_t)acc, 0u));
            i_l = (uint32_t)i - before;
            v_l = d & (0xFFFFFFFFu >> __builtin_clz(i_l));
            const unsigned long long acc2 = CW_BALLOT(v_l <= i_l) & valid;
            if (acc2 == acc) break;
            acc = acc2;
        }
        // accepted draws landing on a low position that still holds its token (a position hit once holds a non-token
        // for the rest of this phase, so later hits change nothing: of ~34 low hits per 21x21 shuffle at most 9 matter)
        unsigned long long ev = CW_BALLOT(v_l <= 8u && !((hit >> (v_l & 15u)) & 1u)) & acc;
        while (ev) {                                 // token events, in draw order (two may name the same position)
            const int l = __builtin_ctzll(ev);
            ev &= ev - 1;
            const uint32_t vv = __builtin_amdgcn_readlane(v_l, l);
            const uint32_t il = __builtin_amdgcn_readlane(i_l, l);
            if (!((hit >> vv) & 1u)) v_tok = (lane == vv) ? il : v_tok;   // token vv is final at position il (> 8)
            hit |= 1u << vv;                                            // the non-token from il lands on vv
        }
        i -= __popcll(acc);
        mt.used += take;
    }
    unsigned long long low = 0;                     // nibble q = token at low position q (15 = none)
#pragma unroll
    for (int q = 0; q < 9; q++) low |= (unsigned long long)(((hit >> q) & 1u) ? 15u : (uint32_t)q) << (4 * q);

    // ---- serial tail: the last <= 9 positions, where both ends of a swap can hold tokens
    while (i >= 1) {
        const uint32_t mask = 0xFFFFFFFFu >> __builtin_clz((uint32_t)i);
        const uint32_t v = mt.next() & mask;
        if (v <= (uint32_t)i) {                      // accepted: swap(perm[i], perm[v])
            const uint32_t b = (v <= 8u) ? (uint32_t)((low >> (4u * v)) & 15ull) : 15u;
            const uint32_t a = (i <= 8) ? (uint32_t)((low >> (4u * i)) & 15ull) : 15u;
            if (b != 15u) v_tok = (lane == b) ? (uint32_t)i : v_tok;   // "v_writelane": position i is final now
            if (v <= 8u) low = (low & ~(15ull << (4u * v))) | ((unsigned long long)a << (4u * v));
            i--;
        }
    }
    const uint32_t t0 = (uint32_t)(low & 15ull);
    if (t0 != 15u) v_tok = (lane == t0) ? 0u : v_tok;
    return v_tok;
}

// imagine_obs works on a copy of the 8 object slots held "lane = slot": lane k (< 8) of v_fp / v_fc is slot k's
// cell / code (lanes >= 8: CW_POS_GONE / 0).  Sixteen wave-uniform values would otherwise sit in SGPRs and every
// rank / count over them would be a chain of scalar compares; here a count is one ballot.
// k-th (row-major) cell not in the occupied set {present slots} (+ extra cell if extra >= 0)
__device__ __forceinline__ uint32_t kth_unoccupied(uint32_t v_fp, int extra, uint32_t k)
{
    uint32_t cand = k;
#pragma unroll 1
    for (int it = 0; it < 10; it++) {
        uint32_t cnt = (uint32_t)__popcll(CW_BALLOT(v_fp <= cand));      // GONE / HELD (>= 0xFFFE) never count
        cnt += (extra >= 0 && (uint32_t)extra <= cand) ? 1u : 0u;
        const uint32_t nc = k + cnt;
        if (nc == cand) break;
        cand = nc;
    }
    return cand;
}
__device__ __forceinline__ uint32_t count_code(uint32_t v_fp, uint32_t v_fc, uint32_t want)
{
    return (uint32_t)__popcll(CW_BALLOT(v_fc == want && v_fp < CW_POS_HELD));
}
// among slots whose code == want (and present), the one with rank `which` in cell order (-1: none)
__device__ __forceinline__ int nth_with_code(uint32_t v_fp, uint32_t v_fc, uint32_t want, uint32_t which)
{
    const bool cand = (v_fc == want) && (v_fp < CW_POS_HELD);
    unsigned long long m = CW_BALLOT(cand);
    uint32_t rank = 0;                               // lane k: candidates in a cell before slot k's
    while (m) {
        const int l = __builtin_ctzll(m);
        m &= m - 1;
        rank += (__builtin_amdgcn_readlane(v_fp, l) < v_fp) ? 1u : 0u;
    }
    const unsigned long long sel = CW_BALLOT(cand && rank == which);
    return sel ? __builtin_ctzll(sel) : -1;
}
#define CW_SET_LANE(v, idx, val) v = ((int)lane == (idx)) ? (val) : v

#define CW_RESET_WAVES 4    // waves (= envs in flight) per workgroup
// One env's reset() (ray.py:156-218) by one wavefront; every value in the result is wave-uniform.
struct CwResetOut {
    uint4 init_pos;          // sample_state placement of objects 0..7
    uint32_t init_agent;
    uint4 goal_pos;          // imagine_obs final state
    uint32_t goal_codes, goal_agent;
    uint32_t desired, subset;
    uint32_t draws;          // raw 32-bit draws taken from the env's stream
};

// menu_fn() yields the env's task-menu id; it is called after the MT state's loads are in flight, so a caller that
// still has to fetch the id (cw_reset_kernel: from the header) overlaps that fetch with them.
template <typename MenuFn>
__device__ __forceinline__ CwResetOut reset_env_wave(const CwParams &P, int env, MenuFn menu_fn, uint32_t *lds_mt, int lane)
{
    CW_STAMP(env, 0);
    const CwMtWave::Pending pend = CwMtWave::load_issue(P.mt + (size_t)env * CW_MT_WORDS, P.mt_idx + env, lane);
    const uint32_t menu_id = menu_fn();
    const CwMenuDev M = P.menus[menu_id];
    CwMtWave mt;
    mt.load_commit(lds_mt, pend, lane);
    CW_STAMP(env, 1);

    // task draw, ray.py:169-174
    const uint32_t ntasks = M.stacking ? mt.randint((uint32_t)M.number_of_tasks) + 1u : 1u;
    unsigned long long perm = 0xFEDCBA9876543210ull;             // task_idx = arange(n_selected)
    for (int i = M.n_selected - 1; i >= 1; i--) {                 // RandomState.shuffle
        const uint32_t j = mt.interval((uint32_t)i);
        const unsigned long long ni = (perm >> (4 * i)) & 15ull, nj = (perm >> (4 * j)) & 15ull;
        perm = (perm & ~(15ull << (4 * i))) | (nj << (4 * i));
        perm = (perm & ~(15ull << (4 * j))) | (ni << (4 * j));
    }
    uint32_t desired = 0;
    for (uint32_t q = 0; q < ntasks; q++) {
        const uint32_t idx = (uint32_t)((perm >> (4 * q)) & 15ull);
        desired |= 1u << (uint32_t)((M.sel_bits >> (4 * idx)) & 15ull);
    }

    CW_STAMP(env, 2);
    // placement: sample_state (ray.py:599-628) or a pooled one (ray.py:630-644); lane v < 9 = token v's cell
    uint32_t v_tok;
    if (P.pool_k == 0) {
        v_tok = shuffle_tokens(mt, P.ncell);
    } else {
        const uint32_t pk = mt.randint((uint32_t)P.pool_k);
        const uint16_t *pp = P.pool + ((size_t)env * P.pool_k + pk) * 9;
        v_tok = lane < 9 ? (uint32_t)pp[lane] : 0u;
    }
    uint32_t agent = __builtin_amdgcn_readlane(v_tok, 8);
    uint32_t v_fp = lane < 8 ? v_tok : (uint32_t)CW_POS_GONE;    // tokens 0..7 are objects 0..7 = slots 0..7
    uint32_t v_fc = lane < 8 ? (uint32_t)lane + 1u : 0u;
    uint32_t ip[8];
#pragma unroll
    for (int k = 0; k < 8; k++) ip[k] = __builtin_amdgcn_readlane(v_tok, k);
    const uint4 init_packed = pack_pos(ip);
    const uint32_t init_agent = agent;
    CW_STAMP(env, 3);

    // imagine_obs, ray.py:220-299, on the slot copy; same code order as the reference
    if (desired & (1u << T_MAKEBREAD)) {                          // :226-231 the wheat -> bread
        CW_SET_LANE(v_fc, 7, (uint32_t)BREAD);
    }
    if (desired & (1u << T_EATBREAD)) {                           // :232-237
        const uint32_t which = mt.randint(count_code(v_fp, v_fc, BREAD));
        const int sl = nth_with_code(v_fp, v_fc, BREAD, which);
        CW_SET_LANE(v_fc, sl, (uint32_t)EMPTY);
        CW_SET_LANE(v_fp, sl, (uint32_t)CW_POS_GONE);
    }
    if (desired & (1u << T_CHOPTREE)) {                           // :238-243 the tree -> sticks
        CW_SET_LANE(v_fc, 4, (uint32_t)STICKS);
    }
    if (desired & (1u << T_MOVESTICKS)) {                         // :244-257
        const uint32_t present = (uint32_t)__popcll(CW_BALLOT(v_fp < CW_POS_HELD));
        const uint32_t which_stick = mt.randint(count_code(v_fp, v_fc, STICKS));
        const uint32_t which_spot = mt.randint((uint32_t)P.ncell - present - 1u);   // no object, no agent (:252)
        const int sl = nth_with_code(v_fp, v_fc, STICKS, which_stick);
        const uint32_t to = kth_unoccupied(v_fp, (int)agent, which_spot);
        CW_SET_LANE(v_fp, sl, to);
    }
    if (desired & (1u << T_BUILDHOUSE)) {                         // :258-264
        const uint32_t which = mt.randint(count_code(v_fp, v_fc, STICKS));
        const int sl = nth_with_code(v_fp, v_fc, STICKS, which);
        CW_SET_LANE(v_fc, sl, (uint32_t)HOUSE);
    }
    if (desired & (1u << T_CHOPROCK)) {                           // :265-268
        CW_SET_LANE(v_fc, 3, (uint32_t)EMPTY);
        CW_SET_LANE(v_fp, 3, (uint32_t)CW_POS_GONE);
    }
    if (desired & (1u << T_GOTOHOUSE)) {                          // :269-276
        const uint32_t which = mt.randint(count_code(v_fp, v_fc, HOUSE));
        const int sl = nth_with_code(v_fp, v_fc, HOUSE, which);
        if (sl >= 0) agent = __builtin_amdgcn_readlane(v_fp, sl);
    }
    if (desired & (1u << T_MOVEAXE)) {                            // :277-286 (agent cell allowed, :282)
        const uint32_t present = (uint32_t)__popcll(CW_BALLOT(v_fp < CW_POS_HELD));
        const uint32_t which_spot = mt.randint((uint32_t)P.ncell - present);
        const uint32_t to = kth_unoccupied(v_fp, -1, which_spot);
        CW_SET_LANE(v_fp, 1, to);
    }
    if (desired & (1u << T_MOVEHAMMER)) {                         // :287-297
        const uint32_t present = (uint32_t)__popcll(CW_BALLOT(v_fp < CW_POS_HELD));
        const uint32_t which_spot = mt.randint((uint32_t)P.ncell - present);
        const uint32_t to = kth_unoccupied(v_fp, -1, which_spot);
        CW_SET_LANE(v_fp, 2, to);
    }
    uint32_t fp[8], fc[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        fp[k] = __builtin_amdgcn_readlane(v_fp, k);
        fc[k] = __builtin_amdgcn_readlane(v_fc, k);
    }

    CW_STAMP(env, 4);
    const uint32_t draws = mt.draws();
    mt.store(P.mt + (size_t)env * CW_MT_WORDS, P.mt_idx + env, lane);   // coalesced write-back
    CW_STAMP(env, 5);
    CwResetOut r;
    uint32_t goal_codes = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) goal_codes |= fc[k] << (4 * k);
    r.init_pos = init_packed;
    r.init_agent = init_agent;
    r.goal_pos = pack_pos(fp);
    r.goal_codes = goal_codes;
    r.goal_agent = agent;
    r.desired = desired;
    r.subset = M.reward_subset ? 1u : 0u;
    r.draws = draws;
    return r;
}

// the header of a freshly reset env (hold = 0, achieved = 0 ray.py:176, step_num = 0 ray.py:203)
__device__ __forceinline__ uint4 reset_header(const CwParams &P, const CwResetOut &r, uint32_t menu_id)
{
    const uint32_t ar = __umulhi(r.init_agent, P.div_magic);
    const uint32_t ac = r.init_agent - ar * P.size;
    uint4 h;
    h.x = ar | (ac << 8) | (menu_id << 24);
    h.y = r.desired << 16;
    h.z = (CW_FLAG_RESET | (r.subset ? CW_FLAG_SUBSET : 0u)) << 16;
    h.w = CW_CODES_INITIAL;
    return h;
}
// the cold per-episode records (read only by the render kernels / get_state)
__device__ __forceinline__ void store_episode_records(const CwParams &P, int env, const CwResetOut &r, bool count_episode)
{
    P.goal_pos[env] = r.goal_pos;
    P.goal_codes[env] = r.goal_codes;
    P.goal_agent[env] = (uint16_t)r.goal_agent;
    P.init_pos[env] = r.init_pos;
    P.init_agent[env] = (uint16_t)r.init_agent;
    if (count_episode) P.ep_no[env] += 1;                         // ray.py:200-201
}

// the record of an episode as the look-ahead arrays hold it (refill) / as the episode arrays hold it after a pop
__device__ __forceinline__ void store_next_record(const CwParams &P, int env, int slot, const CwResetOut &r)
{
    const size_t at = (size_t)slot * P.n_envs + env;
    P.nx_init_pos[at] = r.init_pos;
    P.nx_goal_pos[at] = r.goal_pos;
    P.nx_misc[at] = make_uint4(r.init_agent | (r.goal_agent << 16), r.goal_codes, r.desired | (r.subset << 16) | 0x80000000u, r.draws);      // (VALID)
}
// reset() of EVERY env (cw_reset, ray.py:156-218), one wavefront per env: a waiting look-ahead record is taken over, any other env is
// reset here from its stream.  (cw_reset then sweeps the three frame arrays and refills the records.)
__global__ __launch_bounds__(CW_RESET_WAVES *CW_WAVE) void cw_reset_kernel(CwParams P)
{
    __shared__ uint32_t s_mt[CW_RESET_WAVES][CW_MT_WORDS];
    const int lane = threadIdx.x & (CW_WAVE - 1);
    const int wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x / CW_WAVE);
    const int n_waves = (int)gridDim.x * CW_RESET_WAVES;
    for (int env = (int)blockIdx.x * CW_RESET_WAVES + wave_in_block; env < P.n_envs; env += n_waves) {
        const uint4 v_h = P.hdr[env];                                 // in flight beside the MT state
        const uint32_t menu_id = __builtin_amdgcn_readfirstlane(v_h.x) >> 24;
        const bool count_episode = (__builtin_amdgcn_readfirstlane(v_h.z) & 0xFFFFu) != 0;   // ray.py:200-201
        const uint32_t ctl = P.lookahead ? __builtin_amdgcn_readfirstlane(P.nx_ctl[env]) : 0u;
        if (P.lookahead && (__builtin_amdgcn_readfirstlane(P.nx_misc[(size_t)(ctl & 0xFFu) * P.n_envs + env].z) >> 31)) {        // the next episode is waiting
            if (lane == 0) {
                uint4 h = v_h;
                uint32_t sp[8];
                pop_next_episode(P, env, ctl, h, sp, count_episode);      // (not listed: cw_reset tops every env's ring up right after this kernel)
                P.pos[env] = pack_pos(sp);
                P.hdr[env] = h;
                P.achieved_out[env] = 0;                              // the mask outputs describe the new episode
                P.desired_out[env] = (uint16_t)(h.y >> 16);           // (an auto-reset leaves them at the finished step's values)
            }
            continue;
        }
        const CwResetOut r = reset_env_wave(P, env, [&]() { return menu_id; }, s_mt[wave_in_block], lane);
        if (lane == 0) {
            store_episode_records(P, env, r, count_episode);
            P.pos[env] = r.init_pos;
            P.hdr[env] = reset_header(P, r, menu_id);
            P.achieved_out[env] = 0;
            P.desired_out[env] = (uint16_t)r.desired;
        }
    }
}

// LOOK-AHEAD refill: the next reset()s of every QUEUED env (all_envs: of every env with a free slot in its ring), run ahead of time from
// the env's stream and parked in the nx_* arrays; the stream is left AFTER that reset (nx_misc.w says by how many draws).  Launched by the
// host every few steps, between steps: thousands of resets side by side at one wave each cost ~5 ns per reset where ~200 of them beside every
// sweep cost the step 25 us.  Same device function as the slow path, same stream order: results cannot differ.
__global__ __launch_bounds__(CW_RESET_WAVES *CW_WAVE) void cw_refill_kernel(CwParams P, int all_envs)
{
    __shared__ uint32_t s_mt[CW_RESET_WAVES][CW_MT_WORDS];
    const int lane = threadIdx.x & (CW_WAVE - 1);
    const int wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x / CW_WAVE);
    const int wave = blockIdx.x * CW_RESET_WAVES + wave_in_block;
    const int n_waves = gridDim.x * CW_RESET_WAVES;
    if (P.la_feedback && blockIdx.x == 0 && threadIdx.x == 0)       // (what the host tunes its refill period by: read whenever, never waited for)
        __hip_atomic_store(P.la_feedback, P.counters[5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    // No list: the kernel SCANS the envs' control words, 16 envs per wave and round (one coalesced load), and tops up the rings of those that are QUEUED
    // (all_envs: of every env -- after cw_reset, a re-seed, a checkpoint load).  A list needed a returning atomic per finishing wave of the step kernel
    // (a ticket: a memory round trip on the step's critical path, and contention on one word); the scan reads 4 bytes per env once per refill.
    enum { CHUNK = 16 };
    for (int base = wave * CHUNK; base < P.n_envs; base += n_waves * CHUNK) {
        const int mine = base + lane;
        const uint32_t v_ctl = (lane < CHUNK && mine < P.n_envs) ? P.nx_ctl[mine] : 0u;
        unsigned long long m = CW_BALLOT(lane < CHUNK && mine < P.n_envs && (all_envs || (v_ctl & CW_CTL_QUEUED)));
        while (m) {
            const int l = __builtin_ctzll(m);
            m &= m - 1;
            const int env = base + l;
            const uint32_t ctl = __builtin_amdgcn_readlane(v_ctl, l);
            const uint32_t v_hx = P.hdr[env].x;
            // every free slot behind the waiting records (they follow the head in stream order) gets the next reset() of the env's stream
            const uint32_t head = ctl & 0xFFu;
            for (uint32_t k = 0; k < CW_LA_DEPTH; k++) {
                const uint32_t slot = head + k >= CW_LA_DEPTH ? head + k - CW_LA_DEPTH : head + k;
                if (__builtin_amdgcn_readfirstlane(P.nx_misc[(size_t)slot * P.n_envs + env].z) >> 31) continue;
                const CwResetOut r = reset_env_wave(P, env, [&]() { return __builtin_amdgcn_readfirstlane(v_hx) >> 24; }, s_mt[wave_in_block], lane);
                if (lane == 0) store_next_record(P, env, (int)slot, r);
            }
            if ((ctl & CW_CTL_QUEUED) && lane == 0) P.nx_ctl[env] = head;      // (served)
        }
    }
}

// T consecutive steps of every env in ONE persistent launch (state-only observation mode): each
// wavefront owns 64 consecutive envs, keeps their state in registers, steps them lane-parallel and,
// when the ballot shows finished envs, resets them one after the other with the whole wave
// (reset_env_wave).  Envs never interact, so no wave ever waits for another: no kernel boundaries,
// no done list, no launch latency between steps.  For scripted / random action streams
// (actions[T][N] known up front); with a policy in the loop use cw_step.
__global__ __launch_bounds__(CW_RESET_WAVES *CW_WAVE) void cw_rollout_kernel(CwParams P, const uint8_t *actions, int T,
                                                                             int32_t *rewards, uint8_t *dones, int epw)
{
    __shared__ uint32_t s_mt[CW_RESET_WAVES][CW_MT_WORDS];
    const int lane = threadIdx.x & (CW_WAVE - 1);
    const int wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x / CW_WAVE);
    const int wave = blockIdx.x * CW_RESET_WAVES + wave_in_block;
    // epw = envs per wave (8..64): resets are serial within a wave, so small batches use more, narrower waves
    const int env0 = wave * epw;
    if (env0 >= P.n_envs) return;
    const int env = env0 + lane;
    const bool live = lane < epw && env < P.n_envs;
    const int e = live ? env : env0;                 // idle lanes shadow a valid env (their results are dropped)
    uint4 h = P.hdr[e];
    uint32_t sp[8];
    unpack_pos(P.pos[e], sp);
    uint4 ip = P.init_pos[e];
    uint32_t ctl = P.lookahead ? P.nx_ctl[e] : 0u;    // (the ring's head; kept in step with the pops below)
    unsigned long long n_done = 0, n_succ = 0, n_inv = 0;
    int reward = -1;
    bool done = false;
    uint32_t achieved = 0, desired = 0, step_num = 0;
    for (int t = 0; t < T; t++) {
        const int a = actions[(size_t)t * P.n_envs + e];
        const CwStepOut o = step_env(P, h, sp, a, [&]() { return ip; });
        reward = o.reward; done = o.done; achieved = o.achieved; desired = o.desired; step_num = o.step_num;
        if (live) {
            if (rewards) rewards[(size_t)t * P.n_envs + env] = o.reward;
            if (dones) dones[(size_t)t * P.n_envs + env] = o.done ? 1 : 0;
        }
        const unsigned long long m_all = CW_BALLOT(live && o.done);
        n_done += __popcll(m_all);
        n_succ += __popcll(CW_BALLOT(live && o.success));
        n_inv += __popcll(CW_BALLOT(live && o.invalid));
        if (m_all && live && o.done) { P.episode_length[env] = (int32_t)o.step_num; P.episode_return[env] = episode_return_of(P, o); }
        bool popped = false;                         // auto-reset: the look-ahead record at the head of the env's ring if one waits ...
        const uint32_t ctl_was = ctl;
        if (live && o.done && P.lookahead) {
            popped = pop_next_episode(P, env, ctl, h, sp, true);
            if (popped) { ip = pack_pos(sp); ctl = ((ctl & 0xFFu) + 1u == CW_LA_DEPTH ? 0u : (ctl & 0xFFu) + 1u) | CW_CTL_QUEUED; }
        }
        const unsigned long long m_pop = CW_BALLOT(popped);
        unsigned long long m = m_all & ~m_pop;
        while (m) {                                  // ... else the slow way, one finished env at a time, whole wave
            const int l = __builtin_ctzll(m);
            m &= m - 1;
            const int env_l = env0 + l;
            const uint32_t menu_id = __builtin_amdgcn_readlane(h.x, l) >> 24;
            const CwResetOut r = reset_env_wave(P, env_l, [&]() { return menu_id; }, s_mt[wave_in_block], lane);
            if (lane == 0) {
                store_episode_records(P, env_l, r, true);              // step_num >= 1 here
                if (P.lookahead) note_slow_reset(P, env_l, __builtin_amdgcn_readlane(ctl_was, l));
            }
            if (lane == l) {
                h = reset_header(P, r, menu_id);
                unpack_pos(r.init_pos, sp);
                ip = r.init_pos;
                ctl |= CW_CTL_QUEUED;                // (note_slow_reset's bit, in this lane's copy of the word)
            }
        }
    }
    if (live) {
        P.hdr[env] = h;
        P.pos[env] = pack_pos(sp);
        P.reward[env] = reward;                      // outputs of the last step, as cw_step leaves them
        P.done[env] = done ? 1 : 0;
        P.achieved_out[env] = (uint16_t)achieved;
        P.desired_out[env] = (uint16_t)desired;
        (void)step_num;
    }
    if (lane == 0) {
        atomicAdd(&P.counters[0], (unsigned long long)min(epw, P.n_envs - env0) * (unsigned long long)T);
        if (n_done) atomicAdd(&P.counters[1], n_done);
        if (n_succ) atomicAdd(&P.counters[2], n_succ);
        if (n_inv) atomicAdd(&P.counters[3], n_inv);
    }
}

// The RESIDENT stepper of the single-env loop (N == 1, host-mapped outputs, no auto-reset; state-only or dirty-cell frames): ONE wavefront
// that stays on the card for a bounded time slice and turns `step()` from "launch a kernel, wait for the stream" (~17 us, the reference's own
// step time) into "store a doorbell word, spin on an answer word".  It polls a word in pinned coherent host memory (system-scope acquire
// loads: one PCIe read each), and for every new sequence number runs the same step_env / render_edit as cw_step_kernel on the env's state,
// which it keeps in registers and writes back every step (so the records in device memory are current whenever it has answered), writes
// the step outputs and the <= 2 repainted cells straight into the host-mapped buffers, and releases the answer word.  It LEAVES -- every
// path of the loop reaches one of these within one poll -- when the host raises `stop`, when no request has come for `idle_ticks`, or when
// its time slice `life_ticks` is used up (100 MHz ticks); the host side relaunches it on demand and re-serves a request that raced with an exit
// (cw_engine.cpp: cw_step_resident).  A process that dies leaves a kernel that idles out within `idle_ticks`.
__global__ __launch_bounds__(CW_WAVE) void cw_resident_kernel(CwParams P, CwResident *R, uint32_t seq0, int paint_dirty,
                                                              unsigned long long idle_ticks, unsigned long long life_ticks)
{
    const int lane = threadIdx.x;
    uint4 h = P.hdr[0];
    uint32_t sp[8];
    unpack_pos(P.pos[0], sp);
    const uint4 ip = P.init_pos[0];
    uint32_t last = seq0;
    const unsigned long long t_start = wall_clock64();
    unsigned long long t_last = t_start;
    uint32_t reason = 0;
    for (;;) {
        const unsigned long long bell = __hip_atomic_load(&R->bell, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
        const uint32_t d = (uint32_t)bell;
        const uint32_t seq = d >> 8;
        if (seq != last) {
            if (lane == 0) {
                const int a = (int)(d & 0x7Fu);                                  // (bit 7: also leave obs_one_hot in P.res_onehot)
                const CwStepOut o = step_env(P, h, sp, a, [&]() { return ip; });
                P.hdr[0] = h;
                P.pos[0] = pack_pos(sp);
                P.reward[0] = o.reward;
                P.done[0] = o.done ? 1 : 0;
                P.achieved_out[0] = (uint16_t)o.achieved;
                P.desired_out[0] = (uint16_t)o.desired;
                if (o.done) { P.episode_length[0] = (int32_t)o.step_num; P.episode_return[0] = episode_return_of(P, o); }
                if (paint_dirty && o.changed) {                                  // render_edit, ray.py:522-557 (as in cw_step_kernel)
                    uint8_t *frame = P.obs;
                    const uint32_t hold = (h.x >> 16) & 0xFFu;
                    const uint32_t acell = (h.x & 0xFFu) * P.size + ((h.x >> 8) & 0xFFu);
                    if (P.raster == 1) {
                        const uint32_t r0 = __umulhi(o.dirty0, P.div_magic);
                        alt_paint_tile(frame, P.size, r0, o.dirty0 - r0 * P.size, code_of(h.w, slot_at(sp, o.dirty0)), o.dirty0 == acell, hold);
                        if (o.dirty1 != 0xFFFFFFFFu) {
                            const uint32_t r1 = __umulhi(o.dirty1, P.div_magic);
                            alt_paint_tile(frame, P.size, r1, o.dirty1 - r1 * P.size, code_of(h.w, slot_at(sp, o.dirty1)), o.dirty1 == acell, hold);
                        }
                        alt_paint_strip(frame, P.size, hold, 0, 1, false);
                    } else {
                        paint_cell(frame, P.size, o.dirty0, code_of(h.w, slot_at(sp, o.dirty0)), o.dirty0 == acell, hold, P.div_magic, o.mark0);
                        if (o.dirty1 != 0xFFFFFFFFu)
                            paint_cell(frame, P.size, o.dirty1, code_of(h.w, slot_at(sp, o.dirty1)), o.dirty1 == acell, hold, P.div_magic, o.mark1);
                    }
                }
                atomicAdd(&P.counters[0], 1ull);
                if (o.done) atomicAdd(&P.counters[1], 1ull);
                if (o.success) atomicAdd(&P.counters[2], 1ull);
                if (o.invalid) atomicAdd(&P.counters[3], 1ull);
            }
            if (P.res_onehot && (d & 0x80u)) {                                   // CraftingWorldEnvOneHot: obs_one_hot itself is the observation
                const uint32_t hx = __builtin_amdgcn_readlane(h.x, 0), codes = __builtin_amdgcn_readlane(h.w, 0);   // (onehot.py:369-371)
                uint32_t bp[8];
#pragma unroll
                for (int k = 0; k < 8; k++) bp[k] = __builtin_amdgcn_readlane(sp[k], 0);
                const uint32_t agent_cell = (hx & 0xFFu) * P.size + ((hx >> 8) & 0xFFu), hold = (hx >> 16) & 0xFFu;
                for (int cell = lane; cell < P.ncell; cell += CW_WAVE) {
                    const uint32_t code = code_of(codes, slot_at(bp, (uint32_t)cell));
                    uint32_t bits = code ? (1u << (code - 1)) : 0u;
                    if ((uint32_t)cell == agent_cell) bits |= (1u << 8) | (hold ? (1u << (8 + hold)) : 0u);
                    u32x3 dd;
                    dd.x = (bits & 1u) | ((bits >> 1 & 1u) << 8) | ((bits >> 2 & 1u) << 16) | ((bits >> 3 & 1u) << 24);
                    dd.y = (bits >> 4 & 1u) | ((bits >> 5 & 1u) << 8) | ((bits >> 6 & 1u) << 16) | ((bits >> 7 & 1u) << 24);
                    dd.z = (bits >> 8 & 1u) | ((bits >> 9 & 1u) << 8) | ((bits >> 10 & 1u) << 16) | ((bits >> 11 & 1u) << 24);
                    *(u32x3_a4 *)(P.res_onehot + 12 * cell) = dd;
                }
            }
            __threadfence_system();                                              // every lane's stores, then (lane 0) the answer
            if (lane == 0) __hip_atomic_store(&R->ack, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            last = seq;
            t_last = wall_clock64();
            continue;
        }
        const unsigned long long now = wall_clock64();
        if (bell >> 32) { reason = 1; break; }
        if (now - t_last > idle_ticks) { reason = 2; break; }
        if (now - t_start > life_ticks) { reason = 3; break; }
    }
    if (lane == 0) __hip_atomic_store(&R->exited, reason | (last << 8), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// step() of every engine that resets by itself, in ONE launch: a wavefront owns `epw` consecutive envs (8..64, one per lane) and steps them;
// a finished env takes its look-ahead record over in its own lane (pop_next_episode), and what needs the whole wave comes after, one
// finished env at a time: the reset of an env that found no record (reset_env_wave, the slow path) and, in the pixel modes, the frames a
// reset changes.  paint 0: state-only.  paint 1 (dirty-cell frames): render_edit of the <= 2 changed cells by the env's lane; a finished
// env's three frames by the wave.  paint 2 (full frames): only INIT_OBS and desired_goal of a finished env -- the observation array is swept
// after this kernel (cw_render_pieces_kernel), every env's frame alike, and nothing runs beside that sweep: rounds 1-3 reset and painted
// beside it, which cost the launch 26 us for ~220 finished envs per step (profiles/r04_lookahead.txt).  keep_terminal_obs: the finished
// episode's last frame, from the lane's registers.  No done list, no second launch waiting on the first.
// the frames a finished env leaves to paint, as jobs in LDS that ALL FOUR waves of the workgroup take in turn: with the episode
// phases spread out ~220 envs finish on every step of 65 536, two frames each, and the kernel ends with its slowest wave -- one that finds three
// finished envs among its 64 paints six frames while its three neighbours wait for nothing (15.8 us against 13 with phases in step, round 4)
struct CwPaintJob { uint4 pos; uint32_t codes, agent_hold_kind, env, pad; };      // agent_hold_kind: agent cell | hold << 16 | kind << 24
enum { CW_JOB_INIT = 0, CW_JOB_GOAL = 1, CW_JOB_TERMINAL = 2 };
// One workgroup's share of a step: wave `wave` (global index) steps the envs [wave * epw, wave * epw + epw), finished envs take their records, the workgroup
// paints the frames a reset changes.
// PAINT (0 state-only, 1 dirty-cell frames, 2 full frames) and TERM (keep_terminal_obs) are compile-time: the state-only engine's kernel holds no job
// queue (24.5 KB of LDS), no barrier and no frame code at all -- it ran at occupancy 4 with 50 SGPR spills for branches it never takes (round 5) --
// and the pixel kernels lose the branches of the other mode.
template <int PAINT, bool TERM>
__device__ __forceinline__ void fused_step(const CwParams &P, const void *actions, int act_dtype, int epw, int wave, uint32_t *s_mt_wave,
                                           CwPaintJob *s_jobs, int *s_njobs, int *s_cnt /* [3]: finished, successes, invalid actions of this workgroup */)
{
    constexpr int paint = PAINT;
    const int lane = threadIdx.x & (CW_WAVE - 1);
    const int wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x / CW_WAVE);
    const int env0 = wave * epw;
    const bool wave_live = env0 < P.n_envs;          // (a wave past the batch takes part in the workgroup's barriers and paints its share)
    const int env = env0 + lane;
    const bool live = wave_live && lane < epw && env < P.n_envs;
    const int e = live ? env : (wave_live ? env0 : 0);      // idle lanes shadow a valid env (their results are dropped)
    int a;
    if (act_dtype == 0) a = ((const int32_t *)actions)[e];
    else if (act_dtype == 1) a = (int)((const long long *)actions)[e];
    else a = ((const uint8_t *)actions)[e];
    uint4 h = P.hdr[e];
    uint32_t sp[8];
    unpack_pos(P.pos[e], sp);
    const uint4 ip = P.init_pos[e];                  // (asked for with the rest: a second memory round trip only for lanes that hold something costs the wave the same)
    const uint32_t ctl = P.lookahead ? P.nx_ctl[e] : 0u;      // (the head of the env's ring of look-ahead records: 4 more bytes in the same round trip)
    if (threadIdx.x == 0) { *s_njobs = 0; s_cnt[0] = s_cnt[1] = s_cnt[2] = 0; }
    __syncthreads();                                 // (behind the loads' issue)
    const CwStepOut o = step_env(P, h, sp, a, [&]() { return ip; });
    const bool done = live && o.done;
    if (live) {
        P.reward[env] = o.reward;
        P.done[env] = o.done ? 1 : 0;
        P.achieved_out[env] = (uint16_t)o.achieved;
        P.desired_out[env] = (uint16_t)o.desired;
        if (o.done) { P.episode_length[env] = (int32_t)o.step_num; P.episode_return[env] = episode_return_of(P, o); }
        if constexpr (PAINT == 1) { if (o.changed && !o.done) paint_changed_cells(P, env, h, sp, o); }      // render_edit, :358 (a finished env is repainted whole below)
    }
    // auto-reset, look-ahead first: a finished env takes its next episode's record over in its own lane ...
    const uint4 h_last = h;                          // (the finished episode's last state: keep_terminal_obs paints it below)
    uint32_t sp_last[8];
    if constexpr (TERM) {
#pragma unroll
        for (int k = 0; k < 8; k++) sp_last[k] = sp[k];
    }
    // The public counters.  One same-address atomic per WAVE serialises in the L2: with ~480 of 65 536 envs finishing per step (a policy that succeeds) 1 400
    // atomics on the counters' cache line cost the kernel 4.4 us of its 12.8, and the waves' tickets for round 5's refill list another 2 (profiles/
    // r06_experiments.txt D).  So the waves add up in LDS and ONE lane per workgroup adds to the counters; no list: the refill kernel scans the envs' QUEUED bits.
    const unsigned long long m_all = CW_BALLOT(done);
    const unsigned long long m_succ = CW_BALLOT(live && o.success);
    const unsigned long long m_inv = CW_BALLOT(live && o.invalid);
    if (lane == 0 && (m_all | m_inv)) {
        if (m_all) atomicAdd(&s_cnt[0], (int)__popcll(m_all));
        if (m_succ) atomicAdd(&s_cnt[1], (int)__popcll(m_succ));
        if (m_inv) atomicAdd(&s_cnt[2], (int)__popcll(m_inv));
    }
    bool popped = false;
    CwGoalState goal;
    goal.pos = make_uint4(0, 0, 0, 0); goal.codes = 0; goal.agent = 0;
    if (done && P.lookahead) popped = pop_next_episode(P, env, ctl, h, sp, true, PAINT != 0 ? &goal : nullptr);
    const unsigned long long m_pop = CW_BALLOT(popped);
    __syncthreads();
    if (threadIdx.x == 0) {
        if (blockIdx.x == 0) atomicAdd(&P.counters[0], (unsigned long long)P.n_envs);
        if (s_cnt[0]) atomicAdd(&P.counters[1], (unsigned long long)s_cnt[0]);
        if (s_cnt[1]) atomicAdd(&P.counters[2], (unsigned long long)s_cnt[1]);
        if (s_cnt[2]) atomicAdd(&P.counters[3], (unsigned long long)s_cnt[2]);
    }
    // ---- the jobs of this wave's finished envs (pixel modes): INIT_OBS (with the observation itself in the dirty-cell mode), desired_goal, and with
    //      keep_terminal_obs the finished episode's last frame.  Envs that took a record: by their own lanes.
    constexpr int jpe = TERM ? 3 : 2;
    int jmine = 0;
    if constexpr (PAINT != 0) {
        int jbase = 0;
        if (m_all) {
            if (lane == 0) jbase = atomicAdd(s_njobs, jpe * __popcll(m_all));
            jbase = __shfl(jbase, 0);
        }
        jmine = jbase + jpe * __popcll(m_all & ((1ull << lane) - 1ull));
        if constexpr (TERM) if (done) {
            CwPaintJob &j = s_jobs[jmine + 2];
            j.pos = pack_pos(sp_last);
            j.codes = h_last.w;
            j.agent_hold_kind = ((h_last.x & 0xFFu) * P.size + ((h_last.x >> 8) & 0xFFu)) | (((h_last.x >> 16) & 0xFFu) << 16) | (CW_JOB_TERMINAL << 24);
            j.env = (uint32_t)env;
        }
        if (popped) {
            CwPaintJob &j0 = s_jobs[jmine], &j1 = s_jobs[jmine + 1];
            j0.pos = pack_pos(sp);                    // (the new episode's slots are its reset-time placement)
            j0.codes = CW_CODES_INITIAL;
            j0.agent_hold_kind = ((h.x & 0xFFu) * P.size + ((h.x >> 8) & 0xFFu)) | (CW_JOB_INIT << 24);
            j0.env = (uint32_t)env;
            j1.pos = goal.pos;
            j1.codes = goal.codes;
            j1.agent_hold_kind = goal.agent | (CW_JOB_GOAL << 24);
            j1.env = (uint32_t)env;
        }
    }
    unsigned long long m = m_all & ~m_pop;           // envs that found no record: reset here, one at a time, by the whole wave (rare)
    while (m) {
        const int l = __builtin_ctzll(m);
        m &= m - 1;
        const int env_l = env0 + l;
        const uint32_t menu_id = __builtin_amdgcn_readlane(h.x, l) >> 24;
        const CwResetOut r = reset_env_wave(P, env_l, [&]() { return menu_id; }, s_mt_wave, lane);
        if (lane == 0) {
            store_episode_records(P, env_l, r, true);                       // step_num >= 1 here
            if (P.lookahead) note_slow_reset(P, env_l, __builtin_amdgcn_readlane(ctl, l));
        }
        if (lane == l) {
            h = reset_header(P, r, menu_id);
            unpack_pos(r.init_pos, sp);
            if constexpr (PAINT != 0) {
                CwPaintJob &j0 = s_jobs[jmine], &j1 = s_jobs[jmine + 1];
                j0.pos = r.init_pos; j0.codes = CW_CODES_INITIAL; j0.agent_hold_kind = r.init_agent | (CW_JOB_INIT << 24); j0.env = (uint32_t)env_l;
                j1.pos = r.goal_pos; j1.codes = r.goal_codes; j1.agent_hold_kind = r.goal_agent | (CW_JOB_GOAL << 24); j1.env = (uint32_t)env_l;
            }
        }
    }
    if (live) {
        P.hdr[env] = h;
        P.pos[env] = pack_pos(sp);
    }
    if constexpr (PAINT == 0) return;
    __syncthreads();
    const int n_jobs = *s_njobs;
    for (int jq = wave_in_block; jq < n_jobs; jq += CW_RESET_WAVES) {
        const CwPaintJob &j = s_jobs[jq];
        uint32_t jp[8];
        unpack_pos(make_uint4(__builtin_amdgcn_readfirstlane(j.pos.x), __builtin_amdgcn_readfirstlane(j.pos.y), __builtin_amdgcn_readfirstlane(j.pos.z),
                              __builtin_amdgcn_readfirstlane(j.pos.w)), jp);
        const uint32_t codes = __builtin_amdgcn_readfirstlane(j.codes), ahk = __builtin_amdgcn_readfirstlane(j.agent_hold_kind);
        const size_t off = (size_t)__builtin_amdgcn_readfirstlane(j.env) * P.frame_bytes;
        const uint32_t kind = ahk >> 24;
        uint8_t *d0, *d1 = nullptr;
        if (kind == CW_JOB_INIT) { d0 = P.init_img + off; if (paint == 1) d1 = P.obs + off; }      // (dirty-cell engines: the frame is persistent, the new episode's first frame is painted here too)
        else if (kind == CW_JOB_GOAL) d0 = P.desired_img + off;
        else d0 = P.terminal_img + off;
        paint_state_frame(P, d0, jp, codes, ahk & 0xFFFFu, (ahk >> 16) & 0xFFu, lane, d1);
    }
}
template <int PAINT, bool TERM>
__global__ __launch_bounds__(CW_RESET_WAVES *CW_WAVE) void cw_step_fused_kernel(CwParams P, const void *actions, int act_dtype, int epw)
{
    __shared__ uint32_t s_mt[CW_RESET_WAVES][CW_MT_WORDS];
    __shared__ CwPaintJob s_jobs[PAINT != 0 ? CW_RESET_WAVES * CW_WAVE * (TERM ? 3 : 2) : 1];      // (state-only: no queue -- one unused entry, optimised away)
    __shared__ int s_njobs;
    __shared__ int s_cnt[3];
    const int wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x / CW_WAVE);
    fused_step<PAINT, TERM>(P, actions, act_dtype, epw, blockIdx.x * CW_RESET_WAVES + wave_in_block, s_mt[wave_in_block], s_jobs, &s_njobs, s_cnt);
}
typedef void (*CwStepFusedKernel)(CwParams, const void *, int, int);
static CwStepFusedKernel cw_step_fused_variant(int paint, bool term)
{
    if (paint == 0) return cw_step_fused_kernel<0, false>;
    if (paint == 1) return term ? cw_step_fused_kernel<1, true> : cw_step_fused_kernel<1, false>;
    return term ? cw_step_fused_kernel<2, true> : cw_step_fused_kernel<2, false>;
}

// generate_fixed_states, ray.py:149-154: K placements per env from the env's stream
__global__ __launch_bounds__(CW_RESET_WAVES *CW_WAVE) void cw_pool_kernel(CwParams P)
{
    __shared__ uint32_t s_mt[CW_RESET_WAVES][CW_MT_WORDS];
    const int lane = threadIdx.x & (CW_WAVE - 1);
    const int wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x / CW_WAVE);
    const int n_waves = gridDim.x * CW_RESET_WAVES;
    for (int env = blockIdx.x * CW_RESET_WAVES + wave_in_block; env < P.n_envs; env += n_waves) {
        CwMtWave mt;
        mt.load(s_mt[wave_in_block], P.mt + (size_t)env * CW_MT_WORDS, P.mt_idx + env, lane);
        for (int k = 0; k < P.pool_k; k++) {
            const uint32_t v_tok = shuffle_tokens(mt, P.ncell);
            uint16_t *pp = P.pool + ((size_t)env * P.pool_k + k) * 9;
            if (lane < 9) pp[lane] = (uint16_t)v_tok;
        }
        mt.store(P.mt + (size_t)env * CW_MT_WORDS, P.mt_idx + env, lane);
    }
}

// ------------------------------------------------------------------------------------ whole arrays
// cw_render into a caller's array of ANY alignment (an array the sweep's 16-byte stores can take is swept): one wave per frame
__global__ __launch_bounds__(256) void cw_render_frames_kernel(CwParams P, uint8_t *out)
{
    const int lane = threadIdx.x & (CW_WAVE - 1);
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) / CW_WAVE;
    const int n_waves = gridDim.x * blockDim.x / CW_WAVE;
    for (int env = wave; env < P.n_envs; env += n_waves) {
        uint32_t sp[8], codes, agent_cell, hold;
        load_state_uniform(P, env, CW_SRC_CURRENT, sp, codes, agent_cell, hold);
        paint_state_frame(P, out + (size_t)env * P.frame_bytes, sp, codes, agent_cell, hold, lane);
    }
}

// ---- a frame array as a sweep of ALIGNED 4-KiB PIECES: the one painter of whole arrays ---------------------------------
// Both rasters' frames are almost all zeros.  An AltObs frame is a zero fill plus <= 19 lit pixels (render_frame_alt); a Ray frame is
// black (COLORS_N[0], ray.py:28) except the 4x4-pixel cells of its <= 8 objects and the agent's 2x2 mark (ray.py:476-486).  So the
// sweep's jobs need not follow the frames, the pixel rows or the cells at all: job j is the j-th aligned 4-KiB piece of the frame array --
// four 1-KiB stores of zeros, every lane one 16-byte chunk, the shape of a plain fill whatever the frame size -- followed by those bytes
// of the lit items of the frames the piece overlaps that fall inside it, two frames at a time (lanes 0-31 hold the items of one frame,
// 32-63 of the next; a wave's stores to one address are performed in program order, so the items land on the zeros):
//   AltObs: lane = pixel (8 objects, agent, held item, 9 flag pixels of the strip), three byte stores;
//   Ray   : lane = (object slot, pixel row of its cell): the 12 bytes of that row as three dwords, then (lanes of slot 0, rows 1 and 2)
//           the agent's mark over whatever is there: bytes 3..8 of its cell's rows 1 and 2 = a byte, a dword, a byte.
// A piece overlaps at most floor(4095 / frame_bytes) + 2 frames: two for frames of 4 KiB and more (grids from 10x10 / AltObs 12x12), up to
// nine for the smallest (4x4 AltObs: 540 bytes).  FPJ = that number rounded up to a power of two is a template parameter: records are
// fetched for FPJ frames per job, and a job paints FPJ / 2 pairs (the loop ends with the piece's last frame) -- one pair and no loop at all
// for the frames of the BASELINE configs.
// Jobs go to the waves in address order, four consecutive ones per workgroup (a CU writes 16 KiB contiguous; profiles/history/r03_wave_order.txt).
// Records (agent, hold, codes, positions of the frames of a job) are fetched a batch of 64 / FPJ jobs ahead, one (job, frame) per lane, and
// handed to the lanes that paint with ds_bpermute -- asked for before the fill's stores, used after them; no other LDS or memory round trip
// and no branch in a job but the pace loops and the stores' own predicates.  `src` says which of an env's three states the array shows.
// The sweep covers envs [env_lo, env_lo + env_n) (a chunk of the batch, cw_piece_chunks): offsets inside a chunk are 32-bit.
#define CW_PIECE 4096u             // (a sharp optimum: 2 / 8 / 16 KiB pieces are 79 / 18-24 / 20-27 % slower, profiles/history/r03_pieces.txt G)
#define CW_PIECE_STORES 4          // 1-KiB stores per piece
#ifndef CW_HEAD_JOBS
#define CW_HEAD_JOBS 64            // the jobs of every wave at a launch's start that run a notch slower (~40 us) ...
#endif
#define CW_BUSY_FINISHED 16ull     // ... two notches after a step on which at least this many envs finished
// the array's last, partial piece: zeros for [a0, a1), 16-byte chunks where a whole chunk fits, single bytes after it
__device__ __attribute__((noinline)) void piece_fill_partial(uint8_t *dst_base, uint32_t a0, uint32_t a1, int lane)
{
    for (int s = 0; s < CW_PIECE_STORES; s++) {
        const uint32_t c = a0 + 1024u * s + 16u * lane;
        if (c + 16u <= a1) *(uint4 *)(dst_base + c) = make_uint4(0, 0, 0, 0);
        else for (uint32_t b = c; b < a1; b++) dst_base[b] = 0;
    }
}
template <int RASTER, int FPJ>
__device__ __forceinline__ void render_pieces(const CwParams &P, uint8_t *frames, int src, int period16, int head_extra16, int env_lo, int env_n)
{
    constexpr int JPB = CW_WAVE / FPJ;                                      // jobs per batch of records
    const int lane = threadIdx.x & (CW_WAVE - 1);
    const int wpb = blockDim.x / CW_WAVE;
    const int wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x / CW_WAVE);
    const int n_waves = (int)gridDim.x * wpb;
    const int wave = (int)blockIdx.x * wpb + wave_in_block;                 // (four consecutive pieces per workgroup)
    const uint32_t S = (uint32_t)P.size, FB = P.frame_bytes, row_bytes = (RASTER == 1 ? 9u : 12u) * S;
    uint8_t *const dst_base = frames + (size_t)env_lo * FB;
    const uint32_t total = (uint32_t)env_n * FB;                             // (cw_piece_chunks: < 2^32)
    const int n_jobs = (int)((total + CW_PIECE - 1u) / CW_PIECE);
    if (wave >= n_jobs) return;
    const int q_mine = (n_jobs + n_waves - 1) / n_waves;
    const uint32_t half = (uint32_t)lane >> 5, pl = (uint32_t)lane & 31u;
    // what lane pl paints.  AltObs: pl = pixel (0..7 object slots, 8 agent, 9 held item, 10..18 flag).  Ray: pl = 4 slot + pixel row.
    const uint32_t slot = RASTER == 1 ? (pl & 7u) : (pl >> 2), dy = pl & 3u;
    // the colour tables, one entry per lane, read with ds_bpermute (a select chain is 24 instruction slots): AltObs CPV_COLORS[k] (altobs.py:26-27),
    // Ray COLORS_N by cell code (ray.py:28-30)
    const uint32_t v_cpv = cpv_color(lane & 15), v_rgb = rgb_of_code((uint32_t)lane & 15u);
    const uint32_t m_slot = (RASTER != 1 || pl < 8u) ? 0xFFFFFFFFu : 0u, m_agent = pl == 8u ? 0xFFFFFFFFu : 0u, m_held = pl == 9u ? 0xFFFFFFFFu : 0u;
    const uint32_t m_flag = (pl >= 10u && pl < 19u) ? 0xFFFFFFFFu : 0u;
    const uint32_t sh_pos = 16u * (slot & 1u), sh_item = 4u * slot;
    const uint32_t m_p0 = (slot & 6u) == 0u ? 0xFFFFFFFFu : 0u, m_p1 = (slot & 6u) == 2u ? 0xFFFFFFFFu : 0u;      // the positions dword of the slot
    const uint32_t m_p2 = (slot & 6u) == 4u ? 0xFFFFFFFFu : 0u, m_p3 = (slot & 6u) == 6u ? 0xFFFFFFFFu : 0u;
    const uint32_t fj = pl >= 10u ? pl - 10u : 0u, fjr = (fj >= 6u) ? 2u : (fj >= 3u) ? 1u : 0u;
    const uint32_t off_flag = (3u * S + fjr) * row_bytes + 9u + 3u * (fj - 3u * fjr);      // AltObs lanes pl 10..18: the strip's nine flag pixels
    const bool marks = slot == 0u && (dy == 1u || dy == 2u);                 // Ray: the lanes that paint the agent's mark
    __builtin_amdgcn_s_setprio(3);
    // A JOB'S INSTRUCTIONS COUNT.  One wave per SIMD issues an instruction every ~5 clocks, and a job of ~170 instructions is ~590 ns of a wave's time:
    // about the period the memory system takes (580 ns at 7.2 TB/s).  A dozen instructions less per job moved the sweep from 0.846 to 0.858 of the
    // peak at that clock and let it follow a faster one (profiles/r04_clock.txt K): hence the colour table, the 24-bit multiplies and the 32-bit clock.
    struct Rec { int f; uint32_t hx, hw, o2; uint4 p; };               // (hx: agent row | col << 8 | hold << 16; o2, Ray: the colour of row 2 of the agent's mark)
    auto fetch = [&](int base) {
        Rec r;
        const int i = base + lane / FPJ;
        const int id = i * n_waves + wave;
        r.f = -1; r.hx = 0; r.hw = 0; r.o2 = 0;
        r.p = make_uint4(0, 0, 0, 0);
        if (i < q_mine && id < n_jobs) {
            const int f = (int)(((uint32_t)id * CW_PIECE) / FB) + (lane % FPJ);     // (frame index inside the chunk)
            if (f < env_n) {
                r.f = f;
                const int e = env_lo + f;
                if (src == CW_SRC_CURRENT) {
                    const uint32_t *h = (const uint32_t *)(P.hdr + e);
                    r.hx = h[0];
                    r.hw = h[3];
                    r.p = P.pos[e];
                } else {                                                      // (INIT_OBS / desired_goal arrays; hx: the agent's CELL for now)
                    r.hx = src == CW_SRC_INIT ? (uint32_t)P.init_agent[e] : (uint32_t)P.goal_agent[e];
                    r.hw = src == CW_SRC_INIT ? CW_CODES_INITIAL : P.goal_codes[e];
                    r.p = src == CW_SRC_INIT ? P.init_pos[e] : P.goal_pos[e];
                }
            }
        }
        // (derived values here, not a batch later: measured -- with the wait for the loads moved to the next batch's start the sweep keeps up with
        // 6.8 TB/s instead of 7.2, profiles/r04_clock.txt)
        if (src == CW_SRC_CURRENT) r.hx &= 0x00FFFFFFu;                       // (the menu id)
        else { const uint32_t ar = __umulhi(r.hx, P.div_magic); r.hx = ar | ((r.hx - ar * S) << 8); }      // nothing is held in those states
        if (RASTER != 1) { const uint32_t hold = (r.hx >> 16) & 0xFFu; r.o2 = hold ? rgb_of_code(hold) : 0x00FFFFFFu; }
        return r;
    };
    Rec nxt = fetch(0);
    // THE CLOCK.  period16 != 0: job k of a wave starts no earlier than t0 + k x period (period16 = the period in 1/16 of a 10-ns tick of the
    // constant 100-MHz clock, s_memrealtime), the waves' t0 spread evenly over one period: the launch's stores leave as ONE smooth stream at a
    // set rate -- bytes per second = waves x 4 KiB / period -- instead of at whatever rate the waves' instruction streams happen to produce.
    uint32_t t_next16 = ((uint32_t)__builtin_amdgcn_s_memrealtime() << 4) + (uint32_t)(((long long)wave * period16) / n_waves);      // (32 bits of it: 2.7 s)
    for (int base = 0; base < q_mine; base += JPB) {
        const Rec cur = nxt;
        if (base + JPB < q_mine) nxt = fetch(base + JPB);
        const int in_batch = min(q_mine - base, JPB);
        for (int k = 0; k < in_batch; k++) {
            const int f0 = __builtin_amdgcn_readlane(cur.f, FPJ * k);         // the piece's first frame
            if (f0 < 0) continue;                                             // (past the last job)
            const uint32_t a0 = (uint32_t)((base + k) * n_waves + wave) * CW_PIECE;
            const uint32_t a1 = min(a0 + CW_PIECE, total);
            const uint32_t win = a1 - a0;                                    // (x - a0 < win: x inside the piece)
            // ---- the records of a pair of the job's frames, from the lanes that fetched them to the lanes that paint (the first pair is asked for
            //      before the fill's stores and sleeps and used after them: the LDS round trip is off the job's critical path)
            //      (Ray raster.  The AltObs sweep keeps them after the fill, where they were when its pace was found: with them hoisted its launches
            //      turn bimodal, 0.133 or 0.15 ms against a steady 0.131 -- profiles/history/r03_alt_sweep.txt C; the job's own delays are part of the pace)
            uint32_t hx = 0, hw = 0, o2 = 0, pd = 0;
            auto gather = [&](int pair) {
                const int from = (int)(((uint32_t)(FPJ * k + 2 * pair) + half) << 2);
                hx = (uint32_t)__builtin_amdgcn_ds_bpermute(from, (int)cur.hx);
                hw = (uint32_t)__builtin_amdgcn_ds_bpermute(from, (int)cur.hw);
                o2 = RASTER != 1 ? (uint32_t)__builtin_amdgcn_ds_bpermute(from, (int)cur.o2) : 0u;
                pd = ((uint32_t)__builtin_amdgcn_ds_bpermute(from, (int)cur.p.x) & m_p0) | ((uint32_t)__builtin_amdgcn_ds_bpermute(from, (int)cur.p.y) & m_p1) |
                     ((uint32_t)__builtin_amdgcn_ds_bpermute(from, (int)cur.p.z) & m_p2) | ((uint32_t)__builtin_amdgcn_ds_bpermute(from, (int)cur.p.w) & m_p3);
            };
            if (RASTER != 1) gather(0);
            if (period16) {                                                   // ---- the job's slot (THE CLOCK above)
                const uint32_t now16 = (uint32_t)__builtin_amdgcn_s_memrealtime() << 4;
                const int wait16 = (int)(t_next16 - now16);
                if (wait16 > 0) {
                    for (int z = (wait16 * 5) >> 8; z > 0; z--) __builtin_amdgcn_s_sleep(1);               // (64 clocks each: a little short of the slot ...)
                    while ((int)(t_next16 - ((uint32_t)__builtin_amdgcn_s_memrealtime() << 4)) > 0) { }    // (... the rest on the clock)
                } else if (wait16 < -period16) t_next16 = now16 - (uint32_t)period16;                      // fell behind by more than a job: the debt is forgiven
                t_next16 += (uint32_t)(period16 + (base + k < CW_HEAD_JOBS ? head_extra16 : 0));      // (THE HEAD, cw_render_pieces_kernel)
            }
            // ---- the fill
            uint8_t *const job = dst_base + a0;
            if (win == CW_PIECE) {
#pragma unroll
                for (int s = 0; s < CW_PIECE_STORES; s++) *(uint4 *)(job + 1024 * s + 16 * lane) = make_uint4(0, 0, 0, 0);
            } else piece_fill_partial(dst_base, a0, a1, lane);
            // ---- the lit items of the piece's frames, a pair at a time (branch-free: colours from select chains / the table register)
#pragma unroll
            for (int pair = 0; pair < FPJ / 2; pair++) {
                if (FPJ > 2 && pair > 0 && ((uint32_t)f0 + 2u * pair >= (uint32_t)env_n || ((uint32_t)f0 + 2u * pair) * FB >= a1)) break;   // (wave-uniform)
                if (RASTER == 1 || pair > 0) gather(pair);
                const uint32_t fi = (uint32_t)f0 + 2u * (uint32_t)pair + half;      // this half's frame
                const uint32_t f_base = __umul24(fi, FB);                             // (24-bit operands throughout: v_mul_u32_u24 is full rate, v_mul_lo_u32 a quarter)
                const bool frame_on = fi < (uint32_t)env_n && f_base < a1;
                const uint32_t hold = (hx >> 16) & 0xFFu;
                if (RASTER == 1) {
                    const uint32_t agent_cell = __umul24(hx & 0xFFu, S) + ((hx >> 8) & 0xFFu);
                    // pl 0..7: object slots; 8: the agent, pixel 8 (altobs.py:536); 9: the held item, on its own object pixel at the agent's cell
                    const uint32_t pos = (((pd >> sh_pos) & 0xFFFFu) & m_slot) | (agent_cell & ~m_slot);
                    const uint32_t item = (((hw >> sh_item) & 15u) & m_slot) | (9u & m_agent) | (hold & m_held);
                    const uint32_t r = __umulhi(pos, P.div_magic), c = pos - __umul24(r, S), kk = item - 1u;
                    const uint32_t k3 = (kk >= 6u) ? 2u : (kk >= 3u) ? 1u : 0u;
                    const uint32_t off_obj = __umul24(3u * r + k3, row_bytes) + 9u * c + 3u * (kk - 3u * k3);
                    const uint32_t col_obj = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((kk & 15u) << 2), (int)v_cpv);      // (a select chain here compiles to branches)
                    const bool is_obj = (m_flag == 0u) && pl < 10u && item != 0 && pos < (uint32_t)P.ncell;
                    const bool is_flag = (m_flag & hold) != 0;                            // the strip's flag (altobs.py:557-559)
                    uint32_t p_off = is_obj ? off_obj : is_flag ? off_flag : 0xFFFFFFFFu;
                    uint32_t p_val = is_obj ? col_obj : 0x00FFFFFFu;
                    // the held item's pixel on top of an object's: one store of the sum, byte-wise modulo 256 (render_frame_alt)
                    const uint32_t held_off = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((((uint32_t)lane & 32u) + 9u) << 2), (int)p_off);
                    const bool twice = pl < 8u && p_off != 0xFFFFFFFFu && p_off == held_off;
                    p_val = twice ? (((2u * (p_val & 0xFFu)) & 0xFFu) | ((2u * (p_val & 0xFF00u)) & 0xFF00u) | ((2u * (p_val & 0xFF0000u)) & 0xFF0000u)) : p_val;
                    const unsigned long long m_twice = CW_BALLOT(twice);
                    p_off = (pl == 9u && ((m_twice >> (32u * half)) & 0xFFFFFFFFull)) ? 0xFFFFFFFFu : p_off;
                    if (frame_on && p_off != 0xFFFFFFFFu) {
                        const uint32_t at = f_base + p_off;
                        if (at - a0 < win) dst_base[at] = (uint8_t)p_val;
                        if (at + 1u - a0 < win) dst_base[at + 1u] = (uint8_t)(p_val >> 8);
                        if (at + 2u - a0 < win) dst_base[at + 2u] = (uint8_t)(p_val >> 16);
                    }
                } else {
                    // the object of this lane's slot: pixel row dy of its cell, 12 bytes R G B R | G B R G | B R G B (ray.py:476-481).  A row wholly
                    // inside the piece leaves as ONE 12-byte store -- one write request to the L2 where three dword stores were three, and the L2's
                    // request rate is what bounds the sweep of small frames (profiles/r04_clock.txt F) -- a row across the piece's edge dword by dword
                    const uint32_t pos = (pd >> sh_pos) & 0xFFFFu, code = (hw >> sh_item) & 15u;
                    const uint32_t rgb = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(code << 2), (int)v_rgb);
                    const uint32_t r = __umulhi(pos, P.div_magic), c = pos - __umul24(r, S);
                    const uint32_t seg = f_base + __umul24(4u * r + dy, row_bytes) + 12u * c;
                    const u32x3 d = cell_row_dwords(rgb);
                    const bool is_obj = frame_on && code != 0 && pos < (uint32_t)P.ncell;
                    const bool in0 = seg - a0 < win, in2 = seg + 8u - a0 < win;
                    if (is_obj && in0 && in2) *(u32x3_a4 *)(dst_base + seg) = d;
                    else if (is_obj) {
                        if (in0) *(uint32_t *)(dst_base + seg) = d.x;
                        if (seg + 4u - a0 < win) *(uint32_t *)(dst_base + seg + 4u) = d.y;
                        if (in2) *(uint32_t *)(dst_base + seg + 8u) = d.z;
                    }
                    // the agent's mark, over the object it stands on or the black floor: pixels 1, 2 of rows 1, 2 of its cell -- white, and in row 2
                    // the colour of what it holds (ray.py:483-486)   (carried by the object's own rows instead, it costs more than it saves: r04_clock.txt F)
                    const uint32_t aseg = f_base + __umul24(4u * (hx & 0xFFu) + dy, row_bytes) + 12u * ((hx >> 8) & 0xFFu);
                    const uint32_t o = dy == 2u ? o2 : 0x00FFFFFFu;
                    const bool is_mark = frame_on && marks;
                    if (is_mark && aseg - a0 < win) dst_base[aseg + 3u] = (uint8_t)o;
                    if (is_mark && aseg + 4u - a0 < win) *(uint32_t *)(dst_base + aseg + 4u) = (o >> 8) | (o << 16);
                    if (is_mark && aseg + 8u - a0 < win) dst_base[aseg + 8u] = (uint8_t)(o >> 16);
                }
            }
        }
    }
}
template <int RASTER, int FPJ>
__global__ __launch_bounds__(256) void cw_render_pieces_kernel(CwParams P, uint8_t *frames, int src, int period16, int period16_head, int period16_busy,
                                                               int chunk, int env_lo, int env_n)
{
    // THE HEAD.  What the write path takes from a launch it does not take from its first microseconds: started at its full rate a 7.7-TB/s sweep
    // falls into the saturated regime (0.78 of the peak) and stays there; with the first CW_HEAD_JOBS jobs of every wave (~40 us) a notch slower it
    // holds 7.7 for the rest (0.89).  After a step on which envs finished -- the step kernel has just written their INIT_OBS / desired_goal frames:
    // the steady state of a policy that finishes episodes, ~220 envs, 9 MB, on EVERY step -- the head has to be slower still (profiles/r04_clock.txt
    // K-M).  The sweep of the observation array sees for itself what kind of step it follows: the engine's count of finished envs against what the
    // last such sweep saw.  (chunk: bit 0 the step's first launch -- the others follow a sweep and need no head --, bit 1 its last.)
    const unsigned long long done_now = P.counters[1];
    const unsigned long long finished = src == CW_SRC_CURRENT ? done_now - P.counters[4] : (unsigned long long)P.n_envs;
    const bool busy = finished >= CW_BUSY_FINISHED, storm = finished * 8ull >= (unsigned long long)P.n_envs;
    // (after a step on which an eighth of the batch or more finished -- the all-env time-out step wrote two frames per env, 2.8 GB -- and for the
    //  INIT_OBS / desired_goal arrays of a reset the whole launch runs at the busy head's rate: 7.7 TB/s right after that reads 0.34 ms, this 0.22)
    render_pieces<RASTER, FPJ>(P, frames, src, __builtin_amdgcn_readfirstlane(storm ? period16_busy : period16),
                               __builtin_amdgcn_readfirstlane(!storm && (chunk & 1) ? (busy ? period16_busy : period16_head) - period16 : 0), env_lo, env_n);
    if (src == CW_SRC_CURRENT && (chunk & 2) && blockIdx.x == 0 && threadIdx.x == 0) P.counters[4] = done_now;      // (every wave has read it long ago)
}
// ---- SMALL Ray frames (frame_bytes < 4 KiB: grids up to 9x9, the registered craftingworldflat-v3's 8x8 among them): the GATHER painter ------------
// The sweep above writes a piece as zeros and then its lit items ON TOP: every 12-byte cell row is one more write request to the L2, a piece of small
// frames holds dozens of them, and it is the L2's write-REQUEST rate, not bytes, that bounds it (8x8: 0.75 of the HBM peak, 5x5: 0.35 -- DESIGN 4.3).
// Here every lane computes the FINAL content of its own aligned 16-byte chunk and stores it once: a job is still the j-th aligned 4-KiB piece of the
// array, but it leaves as four plain 1-KiB stores -- one request per 64-byte line, no second pass.  Per job:
//   1. the colour table of the frames the piece overlaps (<= NF of them), in LDS, one dword per cell: zeroed, then SCATTERED into -- lane = (frame,
//      object slot) writes COLORS_N[code] at its cell (one object per cell: no two lanes meet), then one lane per frame ORs the agent's flag
//      (1 + what it holds) into bits 24..26 of its cell;
//   2. chunk q of a frame always covers the same bytes of the same <= 2 cells (a cell row is 12 bytes, frames are multiples of 16 bytes, so a chunk
//      starts 0, 4 or 8 bytes into a cell row and never leaves its frame; the second cell may be the first of the next pixel row): a table built once
//      per workgroup holds, per chunk slot, both cells, their pixel rows and that phase.  A lane reads the slot's entry and the two colours, builds
//      each cell's 12-byte row (with the agent's mark where the flag is set and the pixel row is 1 or 2, ray.py:483-486) and picks its four dwords.
// Records are fetched a batch of 64 / NF jobs ahead, one (job, frame) per lane, exactly as in render_pieces.  Same clock, same launch geometry.
#define CW_GATHER_MAX_S 9
template <int NF>
__global__ __launch_bounds__(256) void cw_render_gather_kernel(CwParams P, uint8_t *frames, int src, int period16, int env_lo, int env_n)
{
    constexpr int JPB = CW_WAVE / NF;
    constexpr int COL_WORDS = (NF * CW_GATHER_MAX_S * CW_GATHER_MAX_S + 255) / 256 * 256;     // (the zeroing writes whole 1-KiB rounds)
    __shared__ uint32_t s_col[256 / CW_WAVE][COL_WORDS];
    __shared__ uint32_t s_slot[CW_GATHER_MAX_S * CW_GATHER_MAX_S * 3];                          // chunk slots of one frame: 48 S^2 / 16
    const int lane = threadIdx.x & (CW_WAVE - 1);
    const int wpb = blockDim.x / CW_WAVE;
    const int wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x / CW_WAVE);
    const int n_waves = (int)gridDim.x * wpb;
    const int wave = (int)blockIdx.x * wpb + wave_in_block;
    const uint32_t S = (uint32_t)P.size, ncell = (uint32_t)P.ncell, FB = P.frame_bytes, row_bytes = 12u * S, Q = FB >> 4;
    // ---- the chunk-slot table (the same for every frame): cell A | cell B << 8 | pixel row of A << 16 | of B << 18 | phase << 20
    for (uint32_t q = threadIdx.x; q < Q; q += blockDim.x) {
        const uint32_t o = 16u * q, y = o / row_bytes, x = o - y * row_bytes, c0 = x / 12u, ph = (x - 12u * c0) >> 2;
        const uint32_t cell_a = (y >> 2) * S + c0;
        uint32_t cell_b, dy_b;
        if (c0 + 1u < S) { cell_b = cell_a + 1u; dy_b = y & 3u; }
        else { const uint32_t y2 = min(y + 1u, 4u * S - 1u); cell_b = (y2 >> 2) * S; dy_b = y2 & 3u; }      // (the next pixel row's first cell; a frame's last chunk ends with its last cell: never here)
        s_slot[q] = cell_a | (cell_b << 8) | ((y & 3u) << 16) | (dy_b << 18) | (ph << 20);
    }
    __syncthreads();
    uint8_t *const dst_base = frames + (size_t)env_lo * FB;
    const uint32_t total = (uint32_t)env_n * FB;
    const int n_jobs = (int)((total + CW_PIECE - 1u) / CW_PIECE);
    if (wave >= n_jobs) return;
    const int q_mine = (n_jobs + n_waves - 1) / n_waves;
    uint32_t *const col = s_col[wave_in_block];
    const uint32_t v_rgb = rgb_of_code((uint32_t)lane & 15u);              // COLORS_N by cell code, one entry per lane (ds_bpermute)
    const uint32_t fr = (uint32_t)lane >> 3, slot = (uint32_t)lane & 7u;   // what this lane scatters: (frame of the job, object slot)
    const uint32_t sh_pos = 16u * (slot & 1u), sh_item = 4u * slot;
    const uint32_t m_p0 = (slot & 6u) == 0u ? 0xFFFFFFFFu : 0u, m_p1 = (slot & 6u) == 2u ? 0xFFFFFFFFu : 0u;
    const uint32_t m_p2 = (slot & 6u) == 4u ? 0xFFFFFFFFu : 0u, m_p3 = (slot & 6u) == 6u ? 0xFFFFFFFFu : 0u;
    const uint32_t q_magic = (uint32_t)((1ull << 32) / Q) + 1u;            // x / Q == mulhi(x, magic) for x < 2^16 (a piece holds 256 chunks)
    // most frames a piece overlaps (cw_frames_per_job's count before rounding) x cells, in rounds of 256 dwords: what a job zeroes of its colour table
    const int zero_rounds = __builtin_amdgcn_readfirstlane((int)((((CW_PIECE - 1u) / FB + 2u) * ncell + 255u) / 256u));
    __builtin_amdgcn_s_setprio(3);
    struct Rec { int f; uint32_t hx, hw; uint4 p; };                      // hx: agent row | col << 8 | hold << 16
    auto fetch = [&](int base) {
        Rec r;
        const int i = base + lane / NF;
        const int id = i * n_waves + wave;
        r.f = -1; r.hx = 0; r.hw = 0;
        r.p = make_uint4(0, 0, 0, 0);
        if (i < q_mine && id < n_jobs) {
            const int f = (int)(((uint32_t)id * CW_PIECE) / FB) + (lane % NF);
            if (f < env_n) {
                r.f = f;
                const int e = env_lo + f;
                if (src == CW_SRC_CURRENT) {
                    const uint32_t *h = (const uint32_t *)(P.hdr + e);
                    r.hx = h[0];
                    r.hw = h[3];
                    r.p = P.pos[e];
                } else {
                    r.hx = src == CW_SRC_INIT ? (uint32_t)P.init_agent[e] : (uint32_t)P.goal_agent[e];
                    r.hw = src == CW_SRC_INIT ? CW_CODES_INITIAL : P.goal_codes[e];
                    r.p = src == CW_SRC_INIT ? P.init_pos[e] : P.goal_pos[e];
                }
            }
        }
        if (src == CW_SRC_CURRENT) r.hx &= 0x00FFFFFFu;
        else { const uint32_t ar = __umulhi(r.hx, P.div_magic); r.hx = ar | ((r.hx - ar * S) << 8); }
        return r;
    };
    Rec nxt = fetch(0);
    uint32_t t_next16 = ((uint32_t)__builtin_amdgcn_s_memrealtime() << 4) + (uint32_t)(((long long)wave * period16) / n_waves);
    for (int base = 0; base < q_mine; base += JPB) {
        const Rec cur = nxt;
        if (base + JPB < q_mine) nxt = fetch(base + JPB);
        const int in_batch = min(q_mine - base, JPB);
        for (int k = 0; k < in_batch; k++) {
            const int f0 = __builtin_amdgcn_readlane(cur.f, NF * k);          // the piece's first frame
            if (f0 < 0) continue;
            const uint32_t a0 = (uint32_t)((base + k) * n_waves + wave) * CW_PIECE;
            const uint32_t a1 = min(a0 + CW_PIECE, total);
            // ---- 1. the colour table of the piece's frames
            for (int z = 0; z < zero_rounds; z++) *(uint4 *)(col + 256 * z + 4 * lane) = make_uint4(0, 0, 0, 0);
            {   // (every lane takes part in the permutes -- a ds_bpermute reads zero from a lane that is switched off --, the frame's lanes write)
                const int from = (int)(((uint32_t)(NF * k) + (fr & (uint32_t)(NF - 1))) << 2);
                const uint32_t hx = (uint32_t)__builtin_amdgcn_ds_bpermute(from, (int)cur.hx), hw = (uint32_t)__builtin_amdgcn_ds_bpermute(from, (int)cur.hw);
                const int ff = __builtin_amdgcn_ds_bpermute(from, cur.f);
                const uint32_t pd = ((uint32_t)__builtin_amdgcn_ds_bpermute(from, (int)cur.p.x) & m_p0) | ((uint32_t)__builtin_amdgcn_ds_bpermute(from, (int)cur.p.y) & m_p1) |
                                    ((uint32_t)__builtin_amdgcn_ds_bpermute(from, (int)cur.p.z) & m_p2) | ((uint32_t)__builtin_amdgcn_ds_bpermute(from, (int)cur.p.w) & m_p3);
                const uint32_t pos = (pd >> sh_pos) & 0xFFFFu, code = (hw >> sh_item) & 15u;
                const uint32_t rgb = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(code << 2), (int)v_rgb);
                const bool frame_on = fr < (uint32_t)NF && ff >= 0 && (uint32_t)ff * FB < a1;
                if (frame_on && code != 0 && pos < ncell) col[fr * ncell + pos] = rgb;
                if (frame_on && slot == 0u)                                   // the agent: 1 + what it holds, in bits 24..26 of its cell (after the objects: a wave's LDS operations keep their order)
                    atomicOr(&col[fr * ncell + (hx & 0xFFu) * S + ((hx >> 8) & 0xFFu)], (1u + ((hx >> 16) & 0xFFu)) << 24);
            }
            if (period16) {                                                   // ---- the job's slot on the clock (render_pieces: THE CLOCK)
                const uint32_t now16 = (uint32_t)__builtin_amdgcn_s_memrealtime() << 4;
                const int wait16 = (int)(t_next16 - now16);
                if (wait16 > 0) {
                    for (int z = (wait16 * 5) >> 8; z > 0; z--) __builtin_amdgcn_s_sleep(1);
                    while ((int)(t_next16 - ((uint32_t)__builtin_amdgcn_s_memrealtime() << 4)) > 0) { }
                } else if (wait16 < -period16) t_next16 = now16 - (uint32_t)period16;
                t_next16 += (uint32_t)period16;
            }
            // ---- 2. every lane its four chunks
            const uint32_t q0 = (a0 >> 4) - (uint32_t)f0 * Q;                 // chunk index of the piece's first chunk, counted from frame f0's first
#pragma unroll
            for (int st = 0; st < CW_PIECE_STORES; st++) {
                const uint32_t g = q0 + 64u * st + (uint32_t)lane;            // (< 256 + Q)
                const uint32_t i = __umulhi(g, q_magic), q = g - i * Q;       // frame of the job, chunk slot in it
                const uint32_t e = s_slot[q];
                const uint32_t ca = col[i * ncell + (e & 0xFFu)], cb = col[i * ncell + ((e >> 8) & 0xFFu)];
                u32x3 da = cell_row_dwords(ca & 0x00FFFFFFu), db = cell_row_dwords(cb & 0x00FFFFFFu);
                const uint32_t dya = (e >> 16) & 3u, dyb = (e >> 18) & 3u, fa = ca >> 24, fb = cb >> 24;
                // the agent's mark on pixels 1, 2 of pixel rows 1 and 2 of its cell: white, row 2 in the colour of what it holds (ray.py:483-486)
                const uint32_t ha = fa == 2u ? (110u | (69u << 8) | (39u << 16)) : fa == 3u ? (255u | (105u << 8) | (180u << 16)) : (100u | (100u << 8) | (200u << 16));
                const uint32_t hb = fb == 2u ? (110u | (69u << 8) | (39u << 16)) : fb == 3u ? (255u | (105u << 8) | (180u << 16)) : (100u | (100u << 8) | (200u << 16));
                if (fa != 0u && (dya == 1u || dya == 2u)) da = overlay_dwords(da, (dya == 2u && fa > 1u) ? ha : 0x00FFFFFFu);
                if (fb != 0u && (dyb == 1u || dyb == 2u)) db = overlay_dwords(db, (dyb == 2u && fb > 1u) ? hb : 0x00FFFFFFu);
                const uint32_t ph = e >> 20;
                uint4 out;
                out.x = ph == 0u ? da.x : ph == 1u ? da.y : da.z;
                out.y = ph == 0u ? da.y : ph == 1u ? da.z : db.x;
                out.z = ph == 0u ? da.z : ph == 1u ? db.x : db.y;
                out.w = ph == 0u ? db.x : ph == 1u ? db.y : db.z;
                const uint32_t at = a0 + 1024u * st + 16u * (uint32_t)lane;
                if (at < a1) *(uint4 *)(dst_base + at) = out;               // (frames are multiples of 16 bytes: a chunk is inside the array or past its end)
            }
        }
    }
}
// ------------------------------------------------------------------------------------ exports
// dense grid codes [N][S][S]; one thread per 4 cells
__global__ __launch_bounds__(256) void cw_export_grid_kernel(CwParams P, uint8_t *out)
{
    const size_t total = (size_t)P.n_envs * P.ncell;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += stride) {
        const uint32_t env = (uint32_t)(g / P.ncell);
        const uint32_t cell = (uint32_t)(g - (size_t)env * P.ncell);
        uint32_t sp[8];
        unpack_pos(P.pos[env], sp);
        out[g] = (uint8_t)code_of(P.hdr[env].w, slot_at(sp, cell));
    }
}
// one-hot [N][S][S][12] (observation_vector_space, ray.py:94-98): 0-7 objects, 8 agent, 9-11 hold
// which: 0 the current state, 1 the episode's goal state (imagine_obs' final_state, the OneHot variant's desired_goal,
// onehot.py:310), 2 the state at reset (its init_observation, onehot.py:203); nothing is held in 1 and 2
__global__ __launch_bounds__(256) void cw_export_onehot_kernel(CwParams P, uint8_t *out, int which)
{
    const size_t total = (size_t)P.n_envs * P.ncell;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += stride) {
        const uint32_t env = (uint32_t)(g / P.ncell);
        const uint32_t cell = (uint32_t)(g - (size_t)env * P.ncell);
        const uint4 h = P.hdr[env];
        uint32_t sp[8];
        unpack_pos(which == 1 ? P.goal_pos[env] : which == 2 ? P.init_pos[env] : P.pos[env], sp);
        const uint32_t codes = which == 1 ? P.goal_codes[env] : which == 2 ? CW_CODES_INITIAL : h.w;
        const uint32_t code = code_of(codes, slot_at(sp, cell));
        const uint32_t agent_cell = which == 1 ? (uint32_t)P.goal_agent[env] : which == 2 ? (uint32_t)P.init_agent[env]
                                                                                            : (h.x & 0xFFu) * P.size + ((h.x >> 8) & 0xFFu);
        const uint32_t hold = which ? 0u : (h.x >> 16) & 0xFFu;
        uint32_t bits = code ? (1u << (code - 1)) : 0u;
        if (cell == agent_cell) bits |= (1u << 8) | (hold ? (1u << (8 + hold)) : 0u);
        // 12 bytes of 0/1
        u32x3 d;
        d.x = (bits & 1u) | ((bits >> 1 & 1u) << 8) | ((bits >> 2 & 1u) << 16) | ((bits >> 3 & 1u) << 24);
        d.y = (bits >> 4 & 1u) | ((bits >> 5 & 1u) << 8) | ((bits >> 6 & 1u) << 16) | ((bits >> 7 & 1u) << 24);
        d.z = (bits >> 8 & 1u) | ((bits >> 9 & 1u) << 8) | ((bits >> 10 & 1u) << 16) | ((bits >> 11 & 1u) << 24);
        *(u32x3_a4 *)(out + g * 12) = d;
    }
}

// ------------------------------------------------------------------------------------ render(state) for ANY one-hot state
// render(state=...) of ray.py:442-486 on caller-supplied (S,S,12) one-hot states, whatever they hold (several objects in a cell,
// more than eight objects, hold flags away from the agent): img = sum over object channels of COLORS_N (the reference's tensordot;
// uint16 here, the sums reach 8 x 255), x4 upscale, agent = the FIRST cell (row-major) with channel 8 set: centre 2x2 := 255, and if
// any cell has a hold channel set, row 4r+2 of that centre := COLORS_N[max over cells of (first set hold channel + 1)].  One
// wavefront per state; off the hot path (the engine's own states are sparse slots and take cw_render).
__global__ __launch_bounds__(256) void cw_render_onehot_kernel(const uint8_t *__restrict__ oh, int n_states, int S, uint32_t div_magic,
                                                               uint16_t *__restrict__ out)
{
    const int lane = threadIdx.x & (CW_WAVE - 1);
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) / CW_WAVE;
    const int n_waves = gridDim.x * blockDim.x / CW_WAVE;
    const int ncell = S * S;
    for (int f = wave; f < n_states; f += n_waves) {
        const uint8_t *st = oh + (size_t)f * ncell * 12;
        uint32_t agent = 0xFFFFFFFFu, hold = 0;
        for (int cell = lane; cell < ncell; cell += CW_WAVE) {
            const uint8_t *c = st + 12 * cell;
            if (c[8] == 1) agent = min(agent, (uint32_t)cell);
            const uint32_t hv = c[9] >= c[10] && c[9] >= c[11] ? (c[9] ? 1u : 0u) : (c[10] >= c[11] ? 2u : 3u);   // argmax([0, h9, h10, h11])
            hold = max(hold, hv);
        }
        for (int off = 32; off; off >>= 1) {
            agent = min(agent, (uint32_t)__shfl_xor((int)agent, off));
            hold = max(hold, (uint32_t)__shfl_xor((int)hold, off));
        }
        const uint32_t hold_rgb = rgb_of_code(hold);
        uint16_t *img = out + (size_t)f * ncell * 48;
        const uint32_t row_px = 4u * S;
        for (int cell = lane; cell < ncell; cell += CW_WAVE) {
            const uint8_t *c = st + 12 * cell;
            uint32_t r = 0, g = 0, b = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const uint32_t col = rgb_of_code((uint32_t)k + 1u), v = c[k];
                r += v * (col & 0xFFu); g += v * ((col >> 8) & 0xFFu); b += v * (col >> 16);
            }
            const uint32_t cr = __umulhi((uint32_t)cell, div_magic), cc = (uint32_t)cell - cr * S;
            for (uint32_t dy = 0; dy < 4; dy++)
                for (uint32_t dx = 0; dx < 4; dx++) {
                    uint32_t pr = r, pg = g, pb = b;
                    if ((uint32_t)cell == agent && (dy == 1 || dy == 2) && (dx == 1 || dx == 2)) {
                        pr = pg = pb = 255u;
                        if (dy == 2 && hold) { pr = hold_rgb & 0xFFu; pg = (hold_rgb >> 8) & 0xFFu; pb = hold_rgb >> 16; }
                    }
                    uint16_t *px = img + ((size_t)(4u * cr + dy) * row_px + 4u * cc + dx) * 3;
                    px[0] = (uint16_t)pr; px[1] = (uint16_t)pg; px[2] = (uint16_t)pb;
                }
        }
    }
}

// ... and CraftingWorldEnvAltObs.render(state) (craftingworld_altobs.py:489-560) on ANY one-hot state: pixel k of a cell's 3x3 tile =
// CPV_COLORS[k] x (state[cell][k] + state[cell][9 + k] for k < 3): objects 0..7 and the agent channel, the hold channels added onto
// items 0..2 (:530-533); then 3 more pixel rows, zero except pixels 3..5 = 255 if any cell has a hold channel set (:557-559).
// Frames [3S+3][3S][3] uint16 (the reference's int image: up to 2 x colour).
__global__ __launch_bounds__(256) void cw_render_onehot_alt_kernel(const uint8_t *__restrict__ oh, int n_states, int S, uint32_t div_magic,
                                                                   uint16_t *__restrict__ out)
{
    const int lane = threadIdx.x & (CW_WAVE - 1);
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) / CW_WAVE;
    const int n_waves = gridDim.x * blockDim.x / CW_WAVE;
    const int ncell = S * S;
    const uint32_t row_px = 3u * S;
    for (int f = wave; f < n_states; f += n_waves) {
        const uint8_t *st = oh + (size_t)f * ncell * 12;
        bool held = false;
        for (int cell = lane; cell < ncell; cell += CW_WAVE) {
            const uint8_t *c = st + 12 * cell;
            held = held || c[9] || c[10] || c[11];
        }
        const bool any_held = CW_BALLOT(held) != 0;
        uint16_t *img = out + (size_t)f * (size_t)(3 * S + 3) * row_px * 3;
        for (int cell = lane; cell < ncell; cell += CW_WAVE) {
            const uint8_t *c = st + 12 * cell;
            const uint32_t cr = __umulhi((uint32_t)cell, div_magic), cc = (uint32_t)cell - cr * S;
#pragma unroll
            for (int k = 0; k < 9; k++) {
                const uint32_t cnt = (uint32_t)c[k] + (k < 3 ? (uint32_t)c[9 + k] : 0u), col = cpv_color(k);
                uint16_t *px = img + ((size_t)(3u * cr + k / 3) * row_px + 3u * cc + k % 3) * 3;
                px[0] = (uint16_t)(cnt * (col & 0xFFu)); px[1] = (uint16_t)(cnt * ((col >> 8) & 0xFFu)); px[2] = (uint16_t)(cnt * (col >> 16));
            }
        }
        for (uint32_t j = (uint32_t)lane; j < 3u * row_px; j += CW_WAVE) {
            const uint32_t x = j % row_px;
            const uint16_t v = (any_held && x >= 3u && x < 6u) ? 255 : 0;
            uint16_t *px = img + ((size_t)(3u * S) * row_px + j) * 3;
            px[0] = v; px[1] = v; px[2] = v;
        }
    }
}

// ------------------------------------------------------------------------------------ seeding
// seed() (ray.py:145-147) at batch scale, one lane per env: numpy RandomState(seed) is init_genrand -- 623 dependent
// multiplies, embarrassingly parallel over envs -- and leaves pos = 624; an injected RandomState state (key, pos) is
// already in P.mt / P.mt_idx.  Either way the first `pos` iterations of numpy's in-place twist loop turn the key into
// the engine's consume-and-replace form (cw_mt.h): words < pos next-generation, the rest current.
__global__ __launch_bounds__(256) void cw_seed_kernel(CwParams P, const uint32_t *seeds)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.n_envs) return;
    uint32_t *s = P.mt + (size_t)i * CW_MT_WORDS;
    int pos;
    if (seeds) {
        uint32_t x = seeds[i];
        s[0] = x;
        for (int k = 1; k < CW_MT_WORDS; k++) {
            x = 1812433253u * (x ^ (x >> 30)) + (uint32_t)k;
            s[k] = x;
        }
        pos = CW_MT_WORDS;
    } else {
        pos = min(max(P.mt_idx[i], 0), CW_MT_WORDS);
    }
    uint32_t cur = s[0];
    for (int k = 0; k < pos; k++) {
        const uint32_t nxt = s[k + 1 == CW_MT_WORDS ? 0 : k + 1];          // (word 0 is already next-generation at k = 623)
        const uint32_t far = s[k + 397 >= CW_MT_WORDS ? k + 397 - CW_MT_WORDS : k + 397];
        const uint32_t y = (cur & 0x80000000u) | (nxt & 0x7fffffffu);
        s[k] = far ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        cur = nxt;
    }
    P.mt_idx[i] = pos == CW_MT_WORDS ? 0 : pos;
}

// ------------------------------------------------------------------------------------ launchers
// frames of a job of the sweep (render_pieces): the most frames an aligned 4-KiB piece can overlap, as a power of two
static inline int cw_frames_per_job(uint32_t frame_bytes)
{
    const int most = (int)((CW_PIECE - 1u) / frame_bytes) + 2;
    int fpj = 2;
    while (fpj < most) fpj <<= 1;
    return fpj;                                  // <= 16: the smallest frame is 540 bytes (AltObs 4x4)
}
// The sweep's waves are in step only at the start of a launch; over thousands of rounds they drift apart (0.62-0.74 of the HBM peak at 2^20
// envs in one launch against 0.86 in eight, profiles/history/r03_other_configs.txt).  So a large batch is swept in CHUNKS of consecutive envs, back-to-back
// launches on one stream of at most render_chunk_rounds rounds of 3 KB per wave (~2.8 GB; 32-bit offsets inside a chunk).  Chunks are whole
// multiples of 4 096 envs: every chunk then starts on a 4-KiB boundary of the array (frames are multiples of 16 bytes for the Ray raster, of
// 2 for AltObs) and every wave of a chunk paints the same number of pieces.  -> number of chunks; *per = envs per chunk (the last one may be shorter)
static inline int cw_piece_chunks(const CwParams &P, const CwTuning &tn, int *per)
{
    const long long waves = (long long)tn.n_cu * (256 / CW_WAVE);
    const long long cap = (long long)(tn.render_chunk_rounds > 0 ? tn.render_chunk_rounds : 1 << 20) * waves * 3072;
    const long long bytes = (long long)P.n_envs * P.frame_bytes;
    int n = (int)((bytes + cap - 1) / cap);
    if (n < 1) n = 1;
    *per = (P.n_envs + n - 1) / n;
    if (n > 1 || bytes >= (1ll << 32)) {
        *per = (int)(((long long)*per + 4095) / 4096 * 4096);
        while ((long long)*per * P.frame_bytes >= (1ll << 32) && *per > 4096) *per = (*per / 2 + 4095) / 4096 * 4096;
    }
    return (P.n_envs + *per - 1) / *per;
}
static inline int cw_render_grid(const CwTuning &tn, long long jobs)
{
    // 4 waves per block, persistent grid-stride.  ONE block per CU (1024 waves chip-wide): the HBM
    // write path saturates with few store streams and gets slower with more of them in flight
    // (0.27 ms at 1 block/CU, 0.30 at 2, 0.32 at 4-8: profiles/history/r01_render_sweeps.txt; the instruction-bound sweeps of small frames gain
    // nothing from a second one either: 8x8 0.0468 vs 0.0473 ms, profiles/r04_clock.txt)
    long long blocks = (jobs + 3) / 4;
    if (blocks > tn.n_cu) blocks = tn.n_cu;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}
// Workgroups of a sweep's launch.  ONE per CU for the frames of the BASELINE configs: their sweep follows its clock to the memory's own edge, and more
// store streams in flight only slow it (cw_render_grid).  SMALL frames (under small_frame_bytes: several frames per 4-KiB piece, dozens of items) are
// another regime -- their jobs are bound by instruction issue and LDS / request latency, not by bytes, and a CU hides that with more waves:
// small_blocks_per_cu workgroups per CU (5x5: 0.34 -> 0.45 of the HBM peak with the same painter, 0.53 with the gather painter;
// profiles/r05_experiments.txt D).  The gather painter wants them whatever the batch (5x5 at 262 144 envs: 0.38 / 0.62 / 0.72 with 1 / 2 / 4); the piece
// sweep only while its launch is short -- from ~300 MB on one workgroup per CU is the better again (8x8 at 262 144 envs: 0.81 / 0.75 / 0.77), section K.
static inline bool cw_use_gather(const CwParams &P, const CwTuning &tn) { return tn.gather && P.raster == 0 && P.size <= tn.gather_max_size; }
static inline int cw_sweep_blocks(const CwParams &P, const CwTuning &tn, long long pieces)
{
    const bool small = (int)P.frame_bytes < tn.small_frame_bytes && (cw_use_gather(P, tn) || pieces * (long long)CW_PIECE < tn.small_launch_bytes);
    const int per_cu = small && pieces > 4ll * tn.n_cu * tn.small_blocks_per_cu ? tn.small_blocks_per_cu : 1;
    return cw_render_grid(tn, pieces) * per_cu;
}
typedef void (*CwSweepKernel)(CwParams, uint8_t *, int, int, int, int, int, int, int);
static CwSweepKernel cw_sweep_kernel(int raster, int fpj)
{
    if (raster == 1) return fpj <= 2 ? cw_render_pieces_kernel<1, 2> : fpj <= 4 ? cw_render_pieces_kernel<1, 4> : fpj <= 8 ? cw_render_pieces_kernel<1, 8> : cw_render_pieces_kernel<1, 16>;
    return fpj <= 2 ? cw_render_pieces_kernel<0, 2> : fpj <= 4 ? cw_render_pieces_kernel<0, 4> : fpj <= 8 ? cw_render_pieces_kernel<0, 8> : cw_render_pieces_kernel<0, 16>;
}
// one frame array [N][frame_bytes] (16-byte aligned) painted from the envs' current / reset-time / goal states
static void cw_launch_sweep(const CwParams &P, const CwTuning &tn, uint8_t *frames, int src, hipStream_t st)
{
    int per = 0;
    const int n_chunks = cw_piece_chunks(P, tn, &per);
    const CwSweepKernel k = cw_sweep_kernel(P.raster, cw_frames_per_job(P.frame_bytes));
    for (int c = 0; c < n_chunks; c++) {
        const int env_n = min(per, P.n_envs - c * per);
        const long long pieces = ((long long)env_n * P.frame_bytes + CW_PIECE - 1) / CW_PIECE;
        if (cw_use_gather(P, tn)) {                              // the smallest Ray frames: every 16-byte chunk computed and stored once
            const int grid = cw_sweep_blocks(P, tn, pieces);
            if (cw_frames_per_job(P.frame_bytes) <= 4)
                hipLaunchKernelGGL(cw_render_gather_kernel<4>, dim3(grid), dim3(256), 0, st, P, frames, src, tn.period16, c * per, env_n);
            else
                hipLaunchKernelGGL(cw_render_gather_kernel<8>, dim3(grid), dim3(256), 0, st, P, frames, src, tn.period16, c * per, env_n);
            continue;
        }
        hipLaunchKernelGGL(k, dim3(cw_sweep_blocks(P, tn, pieces)), dim3(256), 0, st, P, frames, src, tn.period16, tn.period16_head, tn.period16_busy, (c == 0 ? 1 : 0) | (c == n_chunks - 1 ? 2 : 0), c * per, env_n);
    }
}

static inline int cw_reset_grid(const CwTuning &tn, int jobs)
{
    // persistent: one wave per env in flight, reset_blocks_per_cu workgroups (x 4 waves) per CU at most
    int blocks = (jobs + CW_RESET_WAVES - 1) / CW_RESET_WAVES;
    if (blocks > tn.n_cu * tn.reset_blocks_per_cu) blocks = tn.n_cu * tn.reset_blocks_per_cu;
    if (blocks < 1) blocks = 1;
    return blocks;
}

// envs per wavefront of the kernels that reset inline (an inline reset occupies the whole wave, one finished env at a time -- rare now that
// finished envs take their look-ahead records): aim for ~1024 waves (one per SIMD) -- 64 envs per wave for large batches, down to 8 for small ones
static int cw_envs_per_wave(int n, int most = 64)
{
    int epw = most;
    while (epw > 8 && (n + epw - 1) / epw < 1024) epw >>= 1;
    return epw;
}

extern "C" {

// One engine step, everything in stream order on `st`: the step kernel (cw_step_fused_kernel for engines that reset by themselves,
// cw_step_kernel for the others) and, in the FULL pixel mode, the sweep of the observation array.
hipError_t cwk_launch_step(const CwParams *P, const CwTuning *T, const void *actions, int act_dtype, int obs_mode,
                           int auto_reset, hipStream_t st, hipEvent_t *ev /* 6 or null */)
{
    const CwTuning &tn = *T;
    const int n = P->n_envs;
    const bool ev_all = ev && obs_mode != 1;                   // (full-frame mode: only the dominant kernel is bracketed -- every event record costs a pipeline bubble)
    if (ev_all) (void)hipEventRecord(ev[0], st);
    if (auto_reset) {
        const int epw = cw_envs_per_wave(n, tn.step_envs_per_wave);
        const int waves = (n + epw - 1) / epw;
        const int paint = obs_mode == 2 ? 1 : obs_mode == 1 ? 2 : 0;
        hipLaunchKernelGGL(cw_step_fused_variant(paint, paint != 0 && P->terminal_img != nullptr), dim3((waves + CW_RESET_WAVES - 1) / CW_RESET_WAVES),
                           dim3(CW_RESET_WAVES * CW_WAVE), 0, st, *P, actions, act_dtype, epw);
    } else {
        hipLaunchKernelGGL(cw_step_kernel, dim3((n + 255) / 256), dim3(256), 0, st, *P, actions, act_dtype, obs_mode == 2 ? 1 : 0);
    }
    if (ev_all) (void)hipEventRecord(ev[1], st);
    if (ev) for (int k = 2; k < 5; k++) (void)hipEventRecord(ev[k], st);
    if (obs_mode == 1) cw_launch_sweep(*P, tn, P->obs, CW_SRC_CURRENT, st);
    if (ev) (void)hipEventRecord(ev[5], st);
    return hipGetLastError();
}

// look-ahead refill (cw_refill_kernel): the QUEUED envs, or every env with a free slot in its ring
hipError_t cwk_launch_refill(const CwParams *P, const CwTuning *T, int all_envs, hipStream_t st)
{
    hipLaunchKernelGGL(cw_refill_kernel, dim3(cw_reset_grid(*T, P->n_envs)), dim3(CW_RESET_WAVES * CW_WAVE), 0, st, *P, all_envs);
    return hipGetLastError();
}

hipError_t cwk_launch_rollout(const CwParams *P, const uint8_t *actions, int T, int32_t *rewards, uint8_t *dones, hipStream_t st)
{
    const int epw = cw_envs_per_wave(P->n_envs);
    const int waves = (P->n_envs + epw - 1) / epw;
    hipLaunchKernelGGL(cw_rollout_kernel, dim3((waves + CW_RESET_WAVES - 1) / CW_RESET_WAVES), dim3(CW_RESET_WAVES * CW_WAVE), 0, st,
                       *P, actions, T, rewards, dones, epw);
    return hipGetLastError();
}

// every env's observation, INIT_OBS and desired_goal arrays from its current, reset-time and goal states: three sweeps
// (after cw_reset the first two show the same pixels; a restored checkpoint's do not)
hipError_t cwk_launch_render_restore(const CwParams *P, const CwTuning *T, hipStream_t st)
{
    cw_launch_sweep(*P, *T, P->obs, CW_SRC_CURRENT, st);
    cw_launch_sweep(*P, *T, P->init_img, CW_SRC_INIT, st);
    cw_launch_sweep(*P, *T, P->desired_img, CW_SRC_GOAL, st);
    return hipGetLastError();
}

hipError_t cwk_launch_reset_all(const CwParams *P, const CwTuning *T, int obs_mode, hipStream_t st)
{
    hipLaunchKernelGGL(cw_reset_kernel, dim3(cw_reset_grid(*T, P->n_envs)), dim3(CW_RESET_WAVES * CW_WAVE), 0, st, *P);
    if (obs_mode != 0) return cwk_launch_render_restore(P, T, st);
    return hipGetLastError();
}

hipError_t cwk_launch_render_onehot(const CwParams *P, const uint8_t *onehot, int n_states, uint16_t *out, hipStream_t st)
{
    int blocks = (n_states + 3) / 4;
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    if (P->raster == 1) hipLaunchKernelGGL(cw_render_onehot_alt_kernel, dim3(blocks), dim3(256), 0, st, onehot, n_states, P->size, P->div_magic, out);
    else hipLaunchKernelGGL(cw_render_onehot_kernel, dim3(blocks), dim3(256), 0, st, onehot, n_states, P->size, P->div_magic, out);
    return hipGetLastError();
}

hipError_t cwk_launch_resident(const CwParams *P, CwResident *R, uint32_t seq0, int paint_dirty, unsigned long long idle_ticks,
                               unsigned long long life_ticks, hipStream_t st)
{
    hipLaunchKernelGGL(cw_resident_kernel, dim3(1), dim3(CW_WAVE), 0, st, *P, R, seq0, paint_dirty, idle_ticks, life_ticks);
    return hipGetLastError();
}

hipError_t cwk_launch_seed(const CwParams *P, const uint32_t *seeds_dev, hipStream_t st)
{
    hipLaunchKernelGGL(cw_seed_kernel, dim3((P->n_envs + 255) / 256), dim3(256), 0, st, *P, seeds_dev);
    return hipGetLastError();
}

hipError_t cwk_launch_pool(const CwParams *P, const CwTuning *T, hipStream_t st)
{
    const int n = P->n_envs;
    hipLaunchKernelGGL(cw_pool_kernel, dim3(cw_reset_grid(*T, n)), dim3(CW_RESET_WAVES * CW_WAVE), 0, st, *P);
    return hipGetLastError();
}

// the shape of a sweep over one frame array as cw_launch_sweep issues it: launches, waves per launch, jobs (4-KiB pieces) per wave of the first launch
void cwk_sweep_shape(const CwParams *P, const CwTuning *T, int *n_chunks, int *waves, int *jobs_per_wave)
{
    int per = 0;
    *n_chunks = cw_piece_chunks(*P, *T, &per);
    const long long pieces = ((long long)min(per, P->n_envs) * P->frame_bytes + CW_PIECE - 1) / CW_PIECE;
    *waves = cw_sweep_blocks(*P, *T, pieces) * (256 / CW_WAVE);
    *jobs_per_wave = (int)((pieces + *waves - 1) / *waves);
}

// the sweep of the observation array exactly as cw_step issues it (cw_create's calibration times this launch)
hipError_t cwk_launch_sweep_calib(const CwParams *P, const CwTuning *T, hipStream_t st)
{
    cw_launch_sweep(*P, *T, P->obs, CW_SRC_CURRENT, st);
    return hipGetLastError();
}

// ~10 us of one idle wave: stands in for the step kernel between two sweeps in cw_create's calibration
__global__ void cw_idle_kernel(int n)
{
    for (int i = 0; i < n; i++) __builtin_amdgcn_s_sleep(127);
}
hipError_t cwk_launch_idle(hipStream_t st)
{
    hipLaunchKernelGGL(cw_idle_kernel, dim3(1), dim3(64), 0, st, 3);
    return hipGetLastError();
}

// cw_render / cw_set_state: the envs' current frames into any array
hipError_t cwk_launch_render_ext(const CwParams *P, const CwTuning *T, uint8_t *out, hipStream_t st)
{
    if (((uintptr_t)out & 15u) == 0) cw_launch_sweep(*P, *T, out, CW_SRC_CURRENT, st);
    else hipLaunchKernelGGL(cw_render_frames_kernel, dim3(cw_render_grid(*T, P->n_envs)), dim3(256), 0, st, *P, out);   // (the sweep's 16-byte stores want an aligned array)
    return hipGetLastError();
}

hipError_t cwk_launch_export(const CwParams *P, const CwTuning *T, uint8_t *out, int onehot, int which, hipStream_t st)
{
    const size_t total = (size_t)P->n_envs * P->ncell;
    const size_t cap = (size_t)T->n_cu * 32;
    int blocks = (int)((total + 255) / 256 < cap ? (total + 255) / 256 : cap);
    if (blocks < 1) blocks = 1;
    if (onehot) hipLaunchKernelGGL(cw_export_onehot_kernel, dim3(blocks), dim3(256), 0, st, *P, out, which);
    else hipLaunchKernelGGL(cw_export_grid_kernel, dim3(blocks), dim3(256), 0, st, *P, out);
    return hipGetLastError();
}

}  // extern "C"
