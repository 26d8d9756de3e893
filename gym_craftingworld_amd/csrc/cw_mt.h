// cw_mt.h -- MT19937 as numpy's legacy RandomState consumes it (ray.py:169,172,236,...,611,636
// call np_random.randint / .shuffle), restructured for one GPU lane per env.
//
// numpy regenerates all 624 state words in one block ("twist") whenever the position reaches
// 624.  A 624-iteration twist in the middle of a divergent per-lane loop would serialize the
// wavefront, so the state is kept in a "consume-and-replace" form instead: when word k is
// consumed, it is immediately replaced by its next-generation value
//     s[k] <- s[(k+397)%624] ^ twist(s[k], s[(k+1)%624])
// which is exactly iteration k of numpy's twist loop executed lazily (that loop walks k upward
// in place, reading s[k], s[k+1] still old and s[k+397] old for k<227 / new otherwise -- the
// same values this form sees).  Every draw therefore costs the same 2 loads + 1 store, control
// flow stays uniform across lanes, and the output stream is bit-identical to numpy's.
// Invariant: words < idx are next-generation, words >= idx current-generation.
// cw_engine.cpp converts numpy (key,pos) states to and from this form on the host.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define CW_MT_WORDS 624

struct CwMt {
    uint32_t *s;      // this env's 624 words
    int k;            // next word index
    uint32_t cur;     // s[k]
    uint32_t nxt;     // s[(k+1)%624]   (prefetched)
    uint32_t far;     // s[(k+397)%624] (prefetched)

    __device__ __forceinline__ void open(uint32_t *state, int idx)
    {
        s = state;
        k = idx;
        cur = s[k];
        prefetch();
    }
    __device__ __forceinline__ void prefetch()
    {
        int k1 = k + 1;   if (k1 >= CW_MT_WORDS) k1 -= CW_MT_WORDS;
        int k397 = k + 397; if (k397 >= CW_MT_WORDS) k397 -= CW_MT_WORDS;
        nxt = s[k1];
        far = s[k397];
    }
    // genrand_uint32
    __device__ __forceinline__ uint32_t next()
    {
        uint32_t y = (cur & 0x80000000u) | (nxt & 0x7fffffffu);
        uint32_t nw = far ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        uint32_t o = cur;
        s[k] = nw;
        cur = nxt;
        k = (k + 1 >= CW_MT_WORDS) ? 0 : k + 1;
        // k==0 after wrap: s[0] was replaced 623 draws ago; nxt (loaded last draw) already is
        // that new value, as numpy's last twist iteration requires.
        prefetch();
        o ^= (o >> 11);
        o ^= (o << 7) & 0x9d2c5680u;
        o ^= (o << 15) & 0xefc60000u;
        o ^= (o >> 18);
        return o;
    }
    // legacy random_interval(max): mask-and-reject; max == 0 draws nothing
    __device__ __forceinline__ uint32_t interval(uint32_t max)
    {
        if (max == 0) return 0;
        uint32_t mask = 0xFFFFFFFFu >> __clz(max);
        uint32_t v;
        do { v = next() & mask; } while (v > max);
        return v;
    }
    // RandomState.randint(n) == interval(n-1)
    __device__ __forceinline__ uint32_t randint(uint32_t n) { return interval(n - 1); }
};
