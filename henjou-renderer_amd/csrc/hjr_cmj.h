// The integer half of the CMJ sample streams (kernel/cmj.h) in the shape the hot path uses it.  Plain C++ (host and device): the
// native test tests/native/cmj_narrow_test.cpp checks every function here against the loop of cmj.h:60-91 / :38-51 spelled out.
//
// cmj.h:60-91 `permute(i, l, p)` is Kensler's hash-based permutation of [0, l).  The renderer only ever calls it with l = 16 and l = 4
// (cmj.h:110-112).  For l = 2^k the mask w is l - 1, the cycle-walk loop body runs exactly once (its result is masked with w, so it is
// < l), and every step of the body — xor, multiplication mod 2^32, `(i & w) >> s` — produces the low k bits of its result from the
// low k bits of its inputs alone.  So the whole function can be evaluated modulo 2^k: the eight 32-bit multiplication constants shrink
// to their low k bits (several become 1 and vanish, -1 becomes a negation), the `(i & w) >> s` terms with s >= k vanish, and the one
// data-dependent multiplier needs a 24-bit multiplication only.  Same values as the 32-bit loop for every (i < l, p).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define HJR_CMJ_FN __host__ __device__ __forceinline__
#else
#define HJR_CMJ_FN static inline
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define HJR_MUL24(a, b) __umul24((a), (b)) /* v_mul_u32_u24: low 32 bits of the product of the low 24 bits */
#else
#define HJR_MUL24(a, b) ((uint32_t)(((a) & 0xffffffu) * ((b) & 0xffffffu)))
#endif

// permute(i, 16, p), i < 16.  Constants mod 16: 0xe170893d -> 13, 0x0929eb3f -> 15 (-1), 0x6935fa69 * 0x74dcb303 -> 9 * 3 -> 11
// (the `(i & w) >> 11` between them is 0), 0x9e501cc3 -> 3, 0xc860a3df -> 15 (-1); `(i & w) >> 4` is 0.
HJR_CMJ_FN uint32_t hjr_cmj_permute16(uint32_t i, uint32_t p)
{
    uint32_t a = HJR_MUL24(i ^ p, 13u);
    a ^= (p >> 16) ^ (p >> 8);
    a = 0u - a;
    a ^= p >> 23;
    a ^= (a & 15u) >> 1;
    a = HJR_MUL24(a, 1u | (p >> 27));
    a = HJR_MUL24(a, 11u);
    a ^= (a & 15u) >> 2;
    a = HJR_MUL24(a, 3u);
    a ^= (a & 15u) >> 2;
    return (p - a) & 15u; // (-a + p) % 16
}
// permute(i, 4, p), i < 4.  Constants mod 4: 0xe170893d -> 1, 0x0929eb3f -> 3 (-1), 0x6935fa69 -> 1, 0x74dcb303, 0x9e501cc3, 0xc860a3df -> 3 each
// (27 = -1 mod 4); of the `(i & w) >> s` terms only s = 1 survives.
HJR_CMJ_FN uint32_t hjr_cmj_permute4(uint32_t i, uint32_t p)
{
    uint32_t a = i ^ p ^ (p >> 16) ^ (p >> 8);
    a = 0u - a;
    a ^= p >> 23;
    a ^= (a & 3u) >> 1;
    a = HJR_MUL24(a, 1u | (p >> 27));
    return (p - a) & 3u; // (-a + p) % 4
}

// xxhash32 of the four words (n_spp / 16, pixel index, depth, seed) (cmj.h:38-51, called at :122) split where the depth enters: the state
// after the first two words is the same for every draw of a path sample.
#define HJR_XXP2 2246822519u
#define HJR_XXP3 3266489917u
#define HJR_XXP4 668265263u
#define HJR_XXP5 374761393u
HJR_CMJ_FN uint32_t hjr_rotl17(uint32_t h) { return (h << 17) | (h >> 15); }
HJR_CMJ_FN uint32_t hjr_xxhash_head(uint32_t px, uint32_t py, uint32_t pw) // words 0, 1 and the seed word
{
    uint32_t h = pw + HJR_XXP5 + px * HJR_XXP3;
    h = HJR_XXP4 * hjr_rotl17(h);
    h += py * HJR_XXP3;
    return HJR_XXP4 * hjr_rotl17(h);
}
HJR_CMJ_FN uint32_t hjr_xxhash_tail(uint32_t head, uint32_t pz) // word 2 and the avalanche
{
    uint32_t h = head + pz * HJR_XXP3;
    h = HJR_XXP4 * hjr_rotl17(h);
    h = HJR_XXP2 * (h ^ (h >> 15));
    h = HJR_XXP3 * (h ^ (h >> 13));
    return h ^ (h >> 16);
}
