// The hot path's mod-2^k evaluation of cmj.h's permute() and its split xxhash32 (henjou-renderer_amd/csrc/hjr_cmj.h) against the
// 32-bit originals spelled out here (Kensler, "Correlated Multi-Jittered Sampling", listing 2, as kernel/cmj.h:60-91 has it; xxhash32 of
// four words as kernel/cmj.h:38-51).  Built as a shared object: the sweep runs in C++, the single-value entry points let the Python
// test compare with the oracle's own functions too.
#include <stdint.h>
#include "../../henjou-renderer_amd/csrc/hjr_cmj.h"

static uint32_t permute_loop(uint32_t i, uint32_t l, uint32_t p)
{
    uint32_t w = l - 1;
    w |= w >> 1; w |= w >> 2; w |= w >> 4; w |= w >> 8; w |= w >> 16;
    do {
        i ^= p; i *= 0xe170893d;
        i ^= p >> 16;
        i ^= (i & w) >> 4;
        i ^= p >> 8; i *= 0x0929eb3f;
        i ^= p >> 23;
        i ^= (i & w) >> 1; i *= 1 | p >> 27;
        i *= 0x6935fa69;
        i ^= (i & w) >> 11; i *= 0x74dcb303;
        i ^= (i & w) >> 2; i *= 0x9e501cc3;
        i ^= (i & w) >> 2; i *= 0xc860a3df;
        i &= w;
        i ^= i >> 5;
    } while (i >= l);
    return (i + p) % l;
}
static uint32_t xxhash_whole(uint32_t px, uint32_t py, uint32_t pz, uint32_t pw)
{
    const uint32_t P2 = 2246822519U, P3 = 3266489917U, P4 = 668265263U, P5 = 374761393U;
    uint32_t h = pw + P5 + px * P3;
    h = P4 * ((h << 17) | (h >> 15));
    h += py * P3;
    h = P4 * ((h << 17) | (h >> 15));
    h += pz * P3;
    h = P4 * ((h << 17) | (h >> 15));
    h = P2 * (h ^ (h >> 15));
    h = P3 * (h ^ (h >> 13));
    return h ^ (h >> 16);
}
static uint64_t rng(uint64_t& s) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }

extern "C" {
uint32_t narrow_permute16(uint32_t i, uint32_t p) { return hjr_cmj_permute16(i, p); }
uint32_t narrow_permute4(uint32_t i, uint32_t p) { return hjr_cmj_permute4(i, p); }
uint32_t split_xxhash(uint32_t px, uint32_t py, uint32_t pz, uint32_t pw) { return hjr_xxhash_tail(hjr_xxhash_head(px, py, pw), pz); }
// n random p (plus a set of structured ones) x every i: number of mismatches
uint64_t sweep(uint64_t n, uint64_t seed)
{
    uint64_t bad = 0, s = seed | 1u;
    for (uint64_t k = 0; k < n + 4096; k++) {
        uint32_t p;
        if (k < 32) p = 1u << k;                       // single bits
        else if (k < 64) p = ~(1u << (k - 32));        // single holes
        else if (k < 4096) p = (uint32_t)(k * 0x01010101u) ^ (uint32_t)(k << 20); // dense low / high patterns
        else p = (uint32_t)(rng(s) >> 16);
        for (uint32_t i = 0; i < 16; i++) bad += hjr_cmj_permute16(i, p) != permute_loop(i, 16, p);
        for (uint32_t i = 0; i < 4; i++) bad += hjr_cmj_permute4(i, p) != permute_loop(i, 4, p);
        const uint32_t a = (uint32_t)rng(s), b = (uint32_t)(rng(s) >> 11), c = (uint32_t)(rng(s) >> 23), d = (uint32_t)(rng(s) >> 5);
        bad += split_xxhash(a, b, c, d) != xxhash_whole(a, b, c, d);
        bad += split_xxhash(p, k & 0xffffu, k & 15u, 0u) != xxhash_whole(p, k & 0xffffu, k & 15u, 0u);
    }
    return bad;
}
// the permutation property itself: for every p of the sweep the l outputs are distinct
uint64_t sweep_bijective(uint64_t n, uint64_t seed)
{
    uint64_t bad = 0, s = seed | 1u;
    for (uint64_t k = 0; k < n; k++) {
        const uint32_t p = (uint32_t)(rng(s) >> 16);
        uint32_t m16 = 0, m4 = 0;
        for (uint32_t i = 0; i < 16; i++) m16 |= 1u << hjr_cmj_permute16(i, p);
        for (uint32_t i = 0; i < 4; i++) m4 |= 1u << hjr_cmj_permute4(i, p);
        bad += (m16 != 0xffffu) + (m4 != 0xfu);
    }
    return bad;
}
}
