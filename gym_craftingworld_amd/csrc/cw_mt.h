// cw_mt.h -- MT19937 as numpy's legacy RandomState consumes it (ray.py:169,172,236,...,611,636
// call np_random.randint / .shuffle), restructured for one GPU WAVEFRONT per env.
//
// numpy regenerates all 624 state words in one block ("twist") whenever the position reaches
// 624.  The engine keeps the state in a "consume-and-replace" form instead: when word k is
// consumed it is replaced at once by its next-generation value
//     s[k] <- s[(k+397)%624] ^ twist(s[k], s[(k+1)%624])
// which is exactly iteration k of numpy's twist loop executed lazily (that loop walks k upward
// in place, reading s[k], s[k+1] still old and s[k+397] old for k<227 / new otherwise -- the
// same values this form sees), so the output stream is bit-identical to numpy's and there is no
// 624-word block phase.  Invariant: words < idx are next-generation, words >= idx current.
// cw_engine.cpp converts numpy (key,pos) states to and from this form on the host.
//
// CwMtWave runs that recurrence 64 words per step on an LDS copy of the state: lane l handles word
// idx+l.  Lanes read their three inputs before any lane writes (one instruction stream, no
// divergence), s[k+l+1] is the neighbour's OLD word as the recurrence wants, and s[k+l+397] is
// never inside the 64-word window being replaced (397 > 63 and 624-397 > 63).  A chunk's tempered
// outputs stay in one VGPR; the serial consumer pulls output j with v_readlane, so all of its
// bookkeeping is wave-uniform.  The unconsumed tail of the last chunk is un-replaced on store().
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define CW_MT_WORDS 624

struct CwMtWave {
    uint32_t *s;        // LDS copy of this env's 624 words (owned by this wave)
    int idx;            // wave-uniform: index of the first word not yet turned into a chunk
    int used;           // wave-uniform: outputs consumed from the current chunk (64 = none left)
    int gens;           // wave-uniform: chunks generated since load (raw draws consumed so far = 64 * gens - (64 - used))
    int lane;
    uint32_t v_out;     // lane l: tempered output of word (chunk_start + l)
    uint32_t v_old;     // lane l: that word's value before replacement

    static __device__ __forceinline__ int wrap(int i) { return i >= CW_MT_WORDS ? i - CW_MT_WORDS : i; }

    // the state comes in two halves so that a caller can put other loads (and the uses of earlier ones) between
    // the issue of these ten and the wait for them: one memory round trip instead of two or three in a row
    struct Pending { uint32_t w[10]; int gidx; };
    static __device__ __forceinline__ Pending load_issue(const uint32_t *g, const int32_t *gidx_p, int lane_)
    {
        Pending q;
#pragma unroll
        for (int k = 0; k < 10; k++) {
            const int j = lane_ + 64 * k;
            q.w[k] = (j < CW_MT_WORDS) ? g[j] : 0u;
        }
        q.gidx = *gidx_p;
        return q;
    }
    __device__ __forceinline__ void load_commit(uint32_t *lds, const Pending &q, int lane_)
    {
        s = lds;
        lane = lane_;
#pragma unroll
        for (int k = 0; k < 10; k++) {
            const int j = lane_ + 64 * k;
            if (j < CW_MT_WORDS) s[j] = q.w[k];
        }
        idx = __builtin_amdgcn_readfirstlane(q.gidx);
        used = 64;
        gens = 0;
        v_out = 0;
        v_old = 0;
    }
    __device__ __forceinline__ void load(uint32_t *lds, const uint32_t *g, const int32_t *gidx_p, int lane_)
    {
        load_commit(lds, load_issue(g, gidx_p, lane_), lane_);
    }
    __device__ __forceinline__ void gen()
    {
        const int p = wrap(idx + lane);
        const uint32_t a = s[p], b = s[wrap(p + 1)], c = s[wrap(p + 397)];
        const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
        v_old = a;
        uint32_t o = a;
        o ^= (o >> 11);
        o ^= (o << 7) & 0x9d2c5680u;
        o ^= (o << 15) & 0xefc60000u;
        o ^= (o >> 18);
        v_out = o;
        s[p] = c ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        idx = wrap(idx + 64);
        used = 0;
        gens++;
    }
    // raw 32-bit draws consumed since load (what cw_get_mt rewinds a look-ahead record by)
    __device__ __forceinline__ uint32_t draws() const { return (uint32_t)(64 * gens - (64 - used)); }
    // genrand_uint32 (wave-uniform result)
    __device__ __forceinline__ uint32_t next()
    {
        if (used == 64) gen();
        const uint32_t r = __builtin_amdgcn_readlane(v_out, __builtin_amdgcn_readfirstlane(used));
        used++;
        return r;
    }
    // legacy random_interval(max): mask-and-reject; max == 0 draws nothing
    __device__ __forceinline__ uint32_t interval(uint32_t max)
    {
        if (max == 0) return 0;
        const uint32_t mask = 0xFFFFFFFFu >> __builtin_clz(max);
        uint32_t v;
        do { v = next() & mask; } while (v > max);
        return v;
    }
    // RandomState.randint(n) == interval(n-1)
    __device__ __forceinline__ uint32_t randint(uint32_t n) { return interval(n - 1); }

    __device__ __forceinline__ void store(uint32_t *g, int32_t *gidx, int lane_)
    {
        if (used < 64) {                              // give back the chunk's unconsumed tail
            int start = idx - 64;
            if (start < 0) start += CW_MT_WORDS;
            if (lane_ >= used) s[wrap(start + lane_)] = v_old;
            idx = wrap(start + used);
        }
        for (int j = lane_; j < CW_MT_WORDS; j += 64) g[j] = s[j];
        if (lane_ == 0) *gidx = idx;
    }
};
