// CMJ sample streams (kernel/cmj.h) and the small sampling / shading-frame helpers of kernel/math.h, restated for gfx950.
#pragma once
#include "hjr_params.hip.h"

// ------------------------------------------------------------------ kernel/cmj.h
struct CMJState { unsigned long long n_spp; uint32_t scramble, depth, image_idx; }; // cmj.h:53-58

HD uint32_t xxhash32_u4(uint32_t px, uint32_t py, uint32_t pz, uint32_t pw) // cmj.h:38-51
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
HD uint32_t cmj_permute(uint32_t i, uint32_t l, uint32_t p) // cmj.h:60-91
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
HD float cmj_randfloat(uint32_t i, uint32_t p) // cmj.h:93-106
{
    i ^= p;
    i ^= i >> 17; i ^= i >> 10; i *= 0xb36534e5;
    i ^= i >> 12; i ^= i >> 21; i *= 0x93fc4795;
    i ^= 0xdf6e307f;
    i ^= i >> 17; i *= 1 | p >> 18;
    return i * (1.0f / 4294967808.0f);
}
HD f2 cmj(uint32_t index, uint32_t scramble) // cmj.h:108-117
{
    index = cmj_permute(index, 16, scramble * 0x51633e2d);
    uint32_t sx = cmj_permute(index % 4, 4, scramble * 0xa511e9b3);
    uint32_t sy = cmj_permute(index / 4, 4, scramble * 0x63d83595);
    float jx = cmj_randfloat(index, scramble * 0xa399d265);
    float jy = cmj_randfloat(index, scramble * 0x711ad6a5);
    f2 r;
    r.x = (index % 4 + (sy + jx) / 4) / 4;
    r.y = (index / 4 + (sx + jy) / 4) / 4;
    return r;
}
HD f2 cmj_2d(CMJState& st) // cmj.h:119-128
{
    const uint32_t index = (uint32_t)(st.n_spp % 16);
    const uint32_t scramble = xxhash32_u4((uint32_t)(st.n_spp / 16), st.image_idx, st.depth, st.scramble);
    f2 r = cmj(index, scramble);
    st.depth++;
    return r;
}
HD float cmj_1d(CMJState& st) { return cmj_2d(st).x; } // cmj.h:130-133

// ------------------------------------------------------------------ kernel/math.h
HD f3 schlick3(f3 F0, f3 w, f3 n) // math.h:26-29
{
    float term1 = 1.0f - dot(w, n);
    return ssub(1.0f, F0) * p_pow5(term1) + F0;
}
HD float schlick_ior(float no, float ni, f3 w, f3 n) // math.h:31-37
{
    float F0 = (no - ni) / (no + ni);
    F0 = F0 * F0;
    float term1 = 1.0f - dot(w, n);
    return F0 + (1.0f - F0) * p_pow5(term1);
}
HD void orthonormal_basis(f3 n, f3& t, f3& b) // math.h:43-51
{
    float sign = copysignf(1.0f, n.z);
    const float a = -1.0f / (sign + n.z);
    const float bb = n.x * n.y * a;
    t = V(1.0f + sign * n.x * n.x * a, sign * bb, -sign * n.x);
    b = V(bb, sign + n.y * n.y * a, -n.y);
}
HD f3 world_to_local(f3 v, f3 t, f3 n, f3 b) { return V(dot(v, t), dot(v, n), dot(v, b)); } // math.h:53-59
HD f3 local_to_world(f3 v, f3 t, f3 n, f3 b) // math.h:61-71
{
    return V(v.x * t.x + v.y * n.x + v.z * b.x, v.x * t.y + v.y * n.y + v.z * b.y, v.x * t.z + v.y * n.z + v.z * b.z);
}
HD float norm2(f3 v) { return v.x * v.x + v.y * v.y + v.z * v.z; } // math.h:88-90
HD bool refract3(f3 v, f3 n, float ior1, float ior2, f3& r) // math.h:92-103
{
    const f3 t_h = (v - n * dot(v, n)) * (-ior1 / ior2);
    if (norm2(t_h) > 1.0f) return false;
    const f3 t_p = n * (-sqrtf(fmaxf(1.0f - norm2(t_h), 0.0f)));
    r = t_h + t_p;
    return true;
}
