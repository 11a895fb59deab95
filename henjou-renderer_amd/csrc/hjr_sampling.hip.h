// CMJ sample streams (kernel/cmj.h) and the small sampling / shading-frame helpers of kernel/math.h, restated for gfx950.
#pragma once
#include "hjr_params.hip.h"
#include "hjr_cmj.h"

// ------------------------------------------------------------------ kernel/cmj.h
// CMJState (cmj.h:53-58) holds {n_spp, scramble (seed), depth, image_idx}; every draw hashes (n_spp / 16, image_idx, depth, seed) and
// uses n_spp % 16 as the index in its 4 x 4 pattern (cmj.h:119-128).  Only the depth changes between the draws of one path sample, so the
// state carried here is the hash after the three constant words (`head`), the index, and the depth.
struct CMJState { uint32_t head, index, depth; };
HD CMJState cmj_state(unsigned long long n_spp, uint32_t seed, uint32_t image_idx, uint32_t depth)
{
    CMJState st;
    st.head = hjr_xxhash_head((uint32_t)(n_spp / 16), image_idx, seed);
    st.index = (uint32_t)(n_spp % 16);
    st.depth = depth;
    return st;
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
HD f2 cmj(uint32_t index, uint32_t scramble) // cmj.h:108-117; permute(i, 16 / 4, p): hjr_cmj.h
{
    index = hjr_cmj_permute16(index, scramble * 0x51633e2d);
    uint32_t sx = hjr_cmj_permute4(index % 4, scramble * 0xa511e9b3);
    uint32_t sy = hjr_cmj_permute4(index / 4, scramble * 0x63d83595);
    float jx = cmj_randfloat(index, scramble * 0xa399d265);
    float jy = cmj_randfloat(index, scramble * 0x711ad6a5);
    f2 r;
    r.x = (index % 4 + (sy + jx) / 4) / 4;
    r.y = (index / 4 + (sx + jy) / 4) / 4;
    return r;
}
HD f2 cmj_2d(CMJState& st) // cmj.h:119-128
{
    const uint32_t scramble = hjr_xxhash_tail(st.head, st.depth);
    f2 r = cmj(st.index, scramble);
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
