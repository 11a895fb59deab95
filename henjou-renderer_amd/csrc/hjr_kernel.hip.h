// The Henjou hot path as one persistent-wavefront HIP megakernel for gfx950 (CDNA4):
//   ray generation -> software BVH2 traversal / triangle test -> BSDF (Disney + thin-film LUT, negative-IOR glass,
//   multiple-scattering GGX) -> next-event-estimation integrator (also Pathtrace / MIS).
//
// Execution model
//   * one lane owns one pixel and runs its `spp` samples in order, so the per-pixel fp32 sum has a fixed order
//     (bitwise independent of scheduling, tile sharding and GPU count);
//   * wavefronts are persistent: a lane whose pixel is finished pulls the next pixel from a global queue with one
//     wave-aggregated atomic (ballot + mbcnt prefix) — the ray queue never drains until the frame is done;
//   * paths are regenerated in place: a lane whose path ended (Russian roulette, miss, light hit, depth cap)
//     starts its next sample in the same loop iteration, so every trace call runs with (nearly) full waves;
//   * the traversal stack is per lane in LDS ([level][lane] -> conflict-free ds_read/ds_write_b32);
//   * nodes / triangles / shading records / materials / lights are 16-byte-record arrays fetched as dwordx4.
//
// Each device function cites the reference lines it restates (paths relative to the reference's include/).
#pragma once
#include <type_traits>

#include "hjr_layout.h"
#include "hjr_math.hip.h"

struct KParams {
    const float4* nodes;
    const float4* tri_geom;
    const float4* tri_shade;
    const uint32_t* tri_inst;
    const float4* materials;
    const float4* lights;
    const uchar4* lut;
    const uchar4* texels;    // RGBA8 atlas of all material textures
    const uint4* tex_desc;   // per texture slot: (texel offset, width, height, srgb)
    const float* srgb_lut;   // 256-entry sRGB -> linear table (host-computed)
    const float4* sky_tex;   // equirect IBL (float4), null = constant sky
    float4* aov_color;
    float4* aov_albedo;
    float4* aov_normal;
    unsigned int* queue_head;
    unsigned long long* stats;
    int lut_w, lut_h;
    int sky_w, sky_h;
    uint32_t n_lights;
    uint32_t width, height, spp, frame, seed, integrator;
    uint32_t tiles_x, n_owned_items; // items = owned tiles * n_chunks * 64
    uint32_t rank, world;
    uint32_t chunk_spp, n_chunks;    // samples per work item, work items per pixel (hjr_chunking)
    uint32_t n_node_f4, n_tri_f4;    // float4 counts of nodes[] / tri_geom[] (LDS staging)
    uint32_t n_mat_f4, n_light_f4;   // float4 counts of materials[] / lights[] (staged behind the triangles in the LDS variant)
    uint32_t stack_depth;            // traversal stack entries per lane (BVH depth + 1)
    uint32_t* stack_spill;           // memory-path kernels: overflow of the short LDS stacks, [level][lane]
    const uint32_t* tile_order;      // owned tiles, expensive first (hjr_classify_tiles_kernel); null = plain round-robin order
    uint32_t* tile_order_w;          // the same buffer, writable (pre-pass kernels)
    uint32_t* tile_class;            // per owned tile: costliest first hit of its pixel centres: 0 background / light, 1 Disney, 2 metallic (msGGX), 3 glass
    uint32_t* tile_count;            // [0..3] tiles per class, [4..7] scatter cursors
    uint32_t n_owned_tiles;
    uint32_t* tile_bucket;           // per owned tile: sort key of the measured-cost order
    uint32_t* tile_cost;             // per owned tile: closest-hit rays traced for it this frame (feeds the next frame's tile order)
    uint32_t* cost_hist;             // [0..63] tiles per cost bucket, [64..127] scatter cursors
    uint32_t cost_div;               // 64 * spp: rays per tile at one ray per sample
    uint32_t spill_stride;           // lanes in the grid
    float4* part_color;              // [n_chunks][height][width] chunk sums when n_chunks > 1
    float4* part_albedo;
    float4* part_normal;
    float cam_pos[3], cam_dir[3], cam_up[3], cam_right[3];
    float cam_f;
    float sky[3]; // scene_sky_default * ibl_intensity
    float ibl_intensity;
};

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

// ------------------------------------------------------------------ surface record: the fields of Payload the BSDFs read
struct Surface { // kernel/Payload.h:12-42
    f3 basecolor;
    float metallic, roughness, sheen, clearcoat, ior;
    bool is_specular, is_thinfilm;
};

// ------------------------------------------------------------------ thin-film LUT: tex2D<float4>(params.lut_texture, u, v), disneyBRDF.h:11-14
// Sampler state from renderer.h:854-898 (uchar4 -> normalised float, linear, wrap, normalised coords); filtering per the
// CUDA programming guide: texel-centre offset, 1.8 fixed-point weights.
HD f3 lut_fetch(const KParams& P, float u, float v)
{
    if (!P.lut || P.lut_w <= 0 || P.lut_h <= 0) return V1(0.0f);
    int w = P.lut_w, h = P.lut_h;
    float x = u * (float)w - 0.5f, y = v * (float)h - 0.5f;
    float fx = floorf(x), fy = floorf(y);
    float ax = floorf((x - fx) * 256.0f + 0.5f) * (1.0f / 256.0f);
    float ay = floorf((y - fy) * 256.0f + 0.5f) * (1.0f / 256.0f);
    int i0 = (int)fx % w; if (i0 < 0) i0 += w;
    int j0 = (int)fy % h; if (j0 < 0) j0 += h;
    int i1 = (i0 + 1) % w, j1 = (j0 + 1) % h;
    uchar4 c00 = P.lut[j0 * w + i0], c10 = P.lut[j0 * w + i1], c01 = P.lut[j1 * w + i0], c11 = P.lut[j1 * w + i1];
    float w00 = (1.0f - ax) * (1.0f - ay), w10 = ax * (1.0f - ay), w01 = (1.0f - ax) * ay, w11 = ax * ay;
    const float k = 1.0f / 255.0f;
    f3 r;
    r.x = w00 * ((float)c00.x * k) + w10 * ((float)c10.x * k) + w01 * ((float)c01.x * k) + w11 * ((float)c11.x * k);
    r.y = w00 * ((float)c00.y * k) + w10 * ((float)c10.y * k) + w01 * ((float)c01.y * k) + w11 * ((float)c11.y * k);
    r.z = w00 * ((float)c00.z * k) + w10 * ((float)c10.z * k) + w01 * ((float)c01.z * k) + w11 * ((float)c11.z * k);
    return r;
}

// ------------------------------------------------------------------ material textures (renderer.h:740-800) and equirect sky (renderer.h:802-851)
// Build-defined sampling (the closest-hit / miss sources are missing): wrap, bilinear with CUDA's 1.8 fixed-point weights,
// sRGB -> linear per texel before filtering for TexType::sRGB; sky (u, v) = (atan2(d.z, d.x) / 2pi + 0.5, acos(d.y) / pi).
struct Bilin { int i0, i1, j0, j1; float w00, w10, w01, w11; };
HD Bilin bilin(float u, float v, int w, int h)
{
    Bilin b;
    float x = u * (float)w - 0.5f, y = v * (float)h - 0.5f;
    float fx = floorf(x), fy = floorf(y);
    float ax = floorf((x - fx) * 256.0f + 0.5f) * (1.0f / 256.0f);
    float ay = floorf((y - fy) * 256.0f + 0.5f) * (1.0f / 256.0f);
    b.i0 = (int)fx % w; if (b.i0 < 0) b.i0 += w;
    b.j0 = (int)fy % h; if (b.j0 < 0) b.j0 += h;
    b.i1 = (b.i0 + 1) % w; b.j1 = (b.j0 + 1) % h;
    b.w00 = (1.0f - ax) * (1.0f - ay); b.w10 = ax * (1.0f - ay); b.w01 = (1.0f - ax) * ay; b.w11 = ax * ay;
    return b;
}
HD f3 tex_fetch(const KParams& P, int slot, float u, float v)
{
    const uint4 d = P.tex_desc[slot];
    const int w = (int)d.y, h = (int)d.z;
    const Bilin b = bilin(u, v, w, h);
    const uchar4* t = P.texels + d.x;
    const uchar4 c00 = t[b.j0 * w + b.i0], c10 = t[b.j0 * w + b.i1], c01 = t[b.j1 * w + b.i0], c11 = t[b.j1 * w + b.i1];
    f3 r;
    if (d.w) {
        const float* L = P.srgb_lut;
        r.x = b.w00 * L[c00.x] + b.w10 * L[c10.x] + b.w01 * L[c01.x] + b.w11 * L[c11.x];
        r.y = b.w00 * L[c00.y] + b.w10 * L[c10.y] + b.w01 * L[c01.y] + b.w11 * L[c11.y];
        r.z = b.w00 * L[c00.z] + b.w10 * L[c10.z] + b.w01 * L[c01.z] + b.w11 * L[c11.z];
    } else {
        const float k = 1.0f / 255.0f;
        r.x = b.w00 * ((float)c00.x * k) + b.w10 * ((float)c10.x * k) + b.w01 * ((float)c01.x * k) + b.w11 * ((float)c11.x * k);
        r.y = b.w00 * ((float)c00.y * k) + b.w10 * ((float)c10.y * k) + b.w01 * ((float)c01.y * k) + b.w11 * ((float)c11.y * k);
        r.z = b.w00 * ((float)c00.z * k) + b.w10 * ((float)c10.z * k) + b.w01 * ((float)c01.z * k) + b.w11 * ((float)c11.z * k);
    }
    return r;
}
HD f3 sky_fetch(const KParams& P, f3 d)
{
    const float u = p_atan2(d.z, d.x) * 0.15915494309189533577f + 0.5f;
    const float v = p_acos(clampf(d.y, -1.0f, 1.0f)) * HJ_INV_PI;
    const int w = P.sky_w, h = P.sky_h;
    const Bilin b = bilin(u, v, w, h);
    const float4 c00 = P.sky_tex[b.j0 * w + b.i0], c10 = P.sky_tex[b.j0 * w + b.i1], c01 = P.sky_tex[b.j1 * w + b.i0], c11 = P.sky_tex[b.j1 * w + b.i1];
    return V(b.w00 * c00.x + b.w10 * c10.x + b.w01 * c01.x + b.w11 * c11.x,
             b.w00 * c00.y + b.w10 * c10.y + b.w01 * c01.y + b.w11 * c11.y,
             b.w00 * c00.z + b.w10 * c10.z + b.w01 * c01.z + b.w11 * c11.z);
}

// ------------------------------------------------------------------ DisneyBRDF (kernel/disneyBRDF.h:16-327)
#define HJ_LOG_CLEARCOAT_ALPHA2 (-13.8155105579642741f) /* logf(0.001f*0.001f): the only argument clearcoat_D ever sees */
#define HJ_CLEARCOAT_ALPHA 0.001f                          /* lerp(0.1f, 0.001f, 1.0f) with math.h:109-111 */

struct Disney {
    f3 basecolor;
    float alpha, metallic, sheen, clearcoat;
    bool is_thinfilm;
};
HD Disney disney_init(const Surface& s) // :165-177
{
    Disney d;
    d.basecolor = s.basecolor;
    d.alpha = clampf(s.roughness * s.roughness, 0.01f, 1.0f);
    d.metallic = s.metallic;
    d.sheen = s.sheen;
    d.clearcoat = s.clearcoat;
    d.is_thinfilm = s.is_thinfilm;
    return d;
}
HD float ggx_D(float a, f3 wm) // :44-48 (same body in BSDFs.h:507-511)
{
    float term1 = wm.x * wm.x / (a * a) + wm.z * wm.z / (a * a) + wm.y * wm.y;
    float term2 = HJ_PI * a * a * term1 * term1;
    return 1.0f / term2;
}
HD float d_Lambda(float a, f3 w) // :58-61
{
    float delta = 1.0f + (a * a * w.x * w.x + a * a * w.z * w.z) / (w.y * w.y);
    return (-1.0f + sqrtf(delta)) * 0.5f;
}
HD float d_G1(float a, f3 w) { return 1.0f / (1.0f + d_Lambda(a, w)); }                             // :50-52
HD float d_G2(float a, f3 wi, f3 wo) { return 1.0f / (1.0f + d_Lambda(a, wi) + d_Lambda(a, wo)); }   // :54-56
HD float d_getPDFDiffuse(f3 wi) { return fabsf(wi.y) * HJ_INV_PI; }                                  // :40-42
HD f3 d_sampleDiffuse(f2 uv, float& pdf) // :30-38
{
    float theta = 0.5f * p_acos(1.0f - 2.0f * uv.x);
    float phi = 2.0f * HJ_PI * uv.y;
    float sinTheta, cosTheta, sp, cp;
    p_sincos(theta, sinTheta, cosTheta);
    p_sincos(phi, sp, cp);
    f3 wi = V(cp * sinTheta, cosTheta, sp * sinTheta);
    pdf = d_getPDFDiffuse(wi);
    return wi;
}
// spherical-cap VNDF sampling (arXiv 2306.05044): disneyBRDF.h:64-80 == BSDFs.h:616-632
HD f3 sample_visible_normal(float alpha, f2 uv, f3 wo)
{
    f3 strech_wo = normalize(V(wo.x * alpha, wo.y, wo.z * alpha));
    float phi = 2.0f * HJ_PI * uv.x;
    float z = fmaf((1.0f - uv.y), (1.0f + strech_wo.y), -strech_wo.y);
    float sinTheta = sqrtf(clampf(1.0f - z * z, 0.0f, 1.0f));
    float sp, cp;
    p_sincos(phi, sp, cp);
    float x = cp * sinTheta;
    float y = sp * sinTheta;
    f3 c = V(x, z, y);
    f3 h = c + strech_wo;
    return normalize(V(h.x * alpha, h.y, h.z * alpha));
}
HD float d_getPDFSpecular(float a, f3 wm, f3 wo) // :88-90
{
    return 0.25f * ggx_D(a, wm) * d_G1(a, wo) * absdot(wo, wm) / (absdot(wm, wo) * fabsf(wo.y));
}
HD float clearcoat_D(f3 wm, float alpha) // :131-139
{
    float alpha2 = alpha * alpha;
    float t = 1.0f + (alpha2 - 1.0f) * wm.y * wm.y;
    return (alpha2 - 1.0f) / (HJ_PI * HJ_LOG_CLEARCOAT_ALPHA2 * t);
}
HD float d_getPDFClearcoat(f3 wm, f3 wo) // :102-104
{
    return clearcoat_D(wm, HJ_CLEARCOAT_ALPHA) * fabsf(wm.y) / (4.0f * fabsf(dot(wm, wo)));
}
HD f3 d_sampleClearcoat(f2 uv, f3 wo, float& pdf) // :93-100
{
    const float ca = HJ_CLEARCOAT_ALPHA;
    float cosineTheta = sqrtf(fmaxf((1.0f - p_pow(ca * ca, 1.0f - uv.x)) / (1.0f - ca * ca), 0.0f));
    float sinTheta = sqrtf(fmaxf(1.0f - cosineTheta * cosineTheta, 0.0f));
    float phi = HJ_PI2 * uv.y;
    float sp, cp;
    p_sincos(phi, sp, cp);
    f3 wm = V(cp * sinTheta, cosineTheta, sp * sinTheta);
    pdf = d_getPDFClearcoat(wm, wo);
    return wm;
}
HD float f_tSchlick(float wn, float F90) // :106-109
{
    float delta = fmaxf(1.0f - wn, 0.0f);
    return 1.0f + (F90 - 1.0f) * delta * delta * delta * delta * delta;
}
HD float clearcoat_Lambda(f3 w, float alpha) // :126-129
{
    float term1 = 1.0f + (alpha * alpha * w.x * w.x + alpha * alpha * w.z * w.z) / (w.y * w.y);
    return 0.5f * (-1.0f + sqrtf(term1));
}
HD f3 disney_eval(const KParams& P, const Disney& d, f3 wo, f3 wi) // :179-235
{
    f3 wm = normalize(wo + wi);
    float dot_wi_n = fabsf(wi.y);
    float dot_wo_n = fabsf(wi.y); // sic (:189)
    float cosine_d = absdot(wi, wm);
    float F_D90 = 0.5f + 2.0f * d.alpha * cosine_d * cosine_d;
    float f_tsi = f_tSchlick(dot_wi_n, F_D90);
    float f_tso = f_tSchlick(dot_wo_n, F_D90);
    f3 f_diffuse = d.basecolor * f_tsi * f_tso * HJ_INV_PI;
    float deltacos = 1.0f / (dot_wi_n + dot_wo_n) - 0.5f;
    f3 f_subsurface = d.basecolor * HJ_INV_PI * 1.25f * (f_tsi * f_tso * deltacos + 0.5f);
    f3 F0 = lerp3(V1(0.08f), d.basecolor, d.metallic);
    if (d.is_thinfilm) { // :213-217
        float thickness = d.basecolor.x;
        float cosine = absdot(wi, wm);
        F0 = lut_fetch(P, thickness, cosine);
    }
    // specular(), :112-120
    f3 f_specular;
    {
        float ggxD = ggx_D(d.alpha, wm);
        float ggxG = d_G2(d.alpha, wi, wo);
        f3 ggxF = schlick3(F0, wo, wm);
        f_specular = (ggxF * 0.25f * ggxD * ggxG) / (fabsf(wo.y) * fabsf(wi.y));
    }
    float delta = fmaxf(1.0f - absdot(wi, wm), 0.0f);
    f3 f_sheen = V1(1.0f) * d.sheen * delta * delta * delta * delta * delta;
    // clearcoat(), :142-150
    f3 f_clearcoat;
    {
        float cD = clearcoat_D(wm, HJ_CLEARCOAT_ALPHA);
        float cG = 1.0f / (1.0f + clearcoat_Lambda(wi, 0.25f) + clearcoat_Lambda(wo, 0.25f));
        f3 cF = schlick3(V1(0.04f), wo, wm);
        f_clearcoat = ((cF * (0.25f * cD * cG)) / (fabsf(wo.y) * fabsf(wi.y))) * 0.25f;
    }
    // m_subsurface is forced to 0 (:170): lerp(f_diffuse, f_subsurface, 0) = f_diffuse + (f_subsurface - f_diffuse) * 0
    return (lerp3(f_diffuse, f_subsurface, 0.0f) + f_sheen) * (1.0f - d.metallic) + f_specular + f_clearcoat * d.clearcoat;
}
HD f3 disney_sample(const KParams& P, const Disney& d, f3 wo, f3& wi, float& pdf, CMJState& st) // :237-307
{
    float diffuseWeight = 1.0f * (1.0f - d.metallic);
    float specularWeight = 0.5f;
    float clearcoatWeight = 0.0f;
    float sumWeight = diffuseWeight + specularWeight + clearcoatWeight;
    float dw = diffuseWeight / sumWeight;
    float sw = specularWeight / sumWeight;
    float cw = clearcoatWeight / sumWeight;
    float select_p = cmj_1d(st);
    float pdf_diffuse = 1.0f, pdf_specular = 1.0f, pdf_clearcoat = 1.0f;
    f2 xi = cmj_2d(st);
    if (select_p < dw) {
        wi = d_sampleDiffuse(xi, pdf_diffuse);
        f3 wm = normalize(wi + wo);
        pdf_specular = d_getPDFSpecular(d.alpha, wm, wo);
        pdf_clearcoat = d_getPDFClearcoat(wm, wo);
    } else if (select_p < dw + sw) {
        f3 wm = sample_visible_normal(d.alpha, xi, wo);
        pdf_specular = d_getPDFSpecular(d.alpha, wm, wo);
        wi = reflect3(-wo, wm);
        pdf_diffuse = d_getPDFDiffuse(wi);
        pdf_clearcoat = d_getPDFClearcoat(wm, wo);
    } else {
        f3 wm = d_sampleClearcoat(xi, wo, pdf_clearcoat);
        wi = reflect3(-wo, wm);
        pdf_diffuse = d_getPDFDiffuse(wi);
        pdf_specular = d_getPDFSpecular(d.alpha, wm, wo);
    }
    pdf = dw * pdf_diffuse + sw * pdf_specular + cw * pdf_clearcoat;
    if (wi.y < 0.0f) { pdf = 1.0f; return V1(0.0f); }
    return disney_eval(P, d, wo, wi);
}
HD float disney_pdf(const Disney& d, f3 wo, f3 wi) // :309-326
{
    float diffuseWeight = 1.0f * (1.0f - d.metallic);
    float specularWeight = 0.5f, clearcoatWeight = 0.0f;
    float sumWeight = diffuseWeight + specularWeight + clearcoatWeight;
    float dw = diffuseWeight / sumWeight, sw = specularWeight / sumWeight;
    f3 wm = normalize(wo + wi);
    return dw * d_getPDFDiffuse(wi) + sw * d_getPDFSpecular(d.alpha, wm, wo);
}

// ------------------------------------------------------------------ MetaMaterialGlass (kernel/BSDFs.h:404-479): negative refractive index
HD f3 metaglass_sample(float ior, f3 wo, f3& wi, float& pdf, CMJState& st)
{
    const f3 rho = V1(1.0f); // BSDFs.h:998
    float ior_o = 1.0f, ior_i = ior, sign = 1.0f;
    f3 lwo = wo, lwi;
    f3 n = V(0, 1, 0);
    if (wo.y < 0.0f) { ior_o = ior; ior_i = 1.0f; lwo.y = -lwo.y; sign = -1.0f; }
    const float fr = schlick_ior(ior_o, ior_i, lwo, n);
    float p = cmj_1d(st);
    f3 t;
    if (p < fr) lwi = reflect3(-lwo, n);
    else if (refract3(lwo, n, ior_o, ior_i, t)) lwi = reflect3(-t, V(0, -1, 0)); // tangential flip (:454)
    else lwi = reflect3(-lwo, n);
    pdf = 1;
    f3 evalbsdf = rho / fabsf(lwi.y);
    wi = lwi;
    wi.y = sign * wi.y;
    return evalbsdf;
}

// ------------------------------------------------------------------ EnagyConservationGGX (kernel/BSDFs.h:483-852): Heitz multiple-scattering walk
HD float ms_C1(float h) { return fminf(1.0f, fmaxf(0.0f, 0.5f * (h + 1.0f))); }        // :494-500
HD float ms_invC1(float U) { return fmaxf(-1.0f, fminf(1.0f, 2.0f * U - 1.0f)); }      // :502-505
HD float ms_Lambda(float a, f3 v) // :525-532 (the -1.0 / 2.0f literals make this a double expression)
{
    if (v.y > 0.9999f) return 0.0f;
    if (v.y < -0.9999f) return -1.0f;
    float delta = 1.0f + (a * a * v.x * v.x + a * a * v.z * v.z) / (v.y * v.y);
    float sg = (v.y > 0.0f) ? 1.0f : -1.0f;
    return (float)((-1.0 + (double)(sg * sqrtf(delta))) / (double)2.0f);
}
HD float ms_G1_Height(float a, f3 wi, float h0) // :551-563
{
    if (wi.y > 0.9999f) return 1.0f;
    if (wi.y <= 0.0f) return 0.0f;
    const float C1_h0 = ms_C1(h0);
    const float Lambda = ms_Lambda(a, wi);
    return p_pow(C1_h0, Lambda);
}
HD float ms_sampleHeight(float a, f3 wr, float hr, float U) // :566-586
{
    if (wr.y > 0.9999f) return HJ_FLT_MAX;
    if (wr.y < -0.9999f) return ms_invC1(U * ms_C1(hr));
    if (fabsf(wr.y) < 0.0001f) return hr;
    const float G_1_ = ms_G1_Height(a, wr, hr);
    if (U > 1.0f - G_1_) return HJ_FLT_MAX;
    return ms_invC1(ms_C1(hr) / p_pow((1.0f - U), 1.0f / ms_Lambda(a, wr)));
}
HD f3 msggx_sampleBSDF(f3 F0, float alpha, f3 wo_in, f3& wi_out, CMJState& st, float& pdf) // :784-819 + :843-851
{
    f3 wr = -wo_in;
    float hr = 1.0f + ms_invC1(0.999f);
    int order = 0;
    f3 weight = V1(1.0f);
    bool early = false;
    f3 early_ret = V1(0.0f);
    for (;;) {
        float U = cmj_1d(st);
        hr = ms_sampleHeight(alpha, wr, hr, U);
        if (hr == HJ_FLT_MAX) break;
        else order++;
        if (order > 5) { wi_out = V(0, 0, 1); early = true; early_ret = V(0, 0, 0); break; }
        // samplePhaseFunction(-wr, state, weight_1), :737-746
        f3 wi = -wr;
        const f2 uv = cmj_2d(st);
        f3 wm = sample_visible_normal(alpha, uv, wi);
        wr = (-wi) + (wm * 2.0f) * dot(wi, wm);
        f3 weight_1 = schlick3(F0, wi, wm);
        weight = weight * weight_1;
        if ((hr != hr) || (wr.z != wr.z)) { early = true; early_ret = V(0, 0, 1); break; } // wi_out untouched (:813-814)
    }
    f3 bsdf;
    if (early) bsdf = early_ret;
    else { wi_out = wr; bsdf = weight; }
    if (wi_out.y < 0.0f || order > 5) return V1(0.0f); // pdf stays as the caller initialised it (:846-848)
    pdf = fabsf(wi_out.y);
    return bsdf;
}

// ------------------------------------------------------------------ BSDF dispatch (kernel/BSDFs.h:979-1038)
HD f3 bsdf_eval(const KParams& P, const Surface& s, f3 wo, f3 wi)
{
    if (s.is_specular) return V1(0.0f);
    Disney d = disney_init(s);
    return disney_eval(P, d, wo, wi);
}
HD f3 bsdf_sample(const KParams& P, const Surface& s, f3 wo, f3& wi, float& pdf, CMJState& st)
{
    if (s.is_specular) return metaglass_sample(s.ior, wo, wi, pdf, st);
    if (!(s.metallic > 0.5f)) {
        Disney d = disney_init(s);
        return disney_sample(P, d, wo, wi, pdf, st);
    }
    return msggx_sampleBSDF(s.basecolor, clampf(s.roughness * s.roughness, 0.0001f, 1.0f), wo, wi, st, pdf);
}
HD float bsdf_pdf(const Surface& s, f3 wo, f3 wi)
{
    if (s.is_specular) return 0.0f;
    Disney d = disney_init(s);
    return disney_pdf(d, wo, wi);
}

// ------------------------------------------------------------------ traversal: replaces optixTrace (kernel/rt.h:15-69) + RT cores
struct Counters {
    uint32_t box, tri;
#ifdef HJR_TIMING
    unsigned long long t_node, t_leaf;
#endif
};

HD float dotf(f3 a, f3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
HD f3 crossf(f3 a, f3 b)
{
    return V(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x)));
}
// canonical ray/triangle test — the same operation sequence as the oracle's ray_tri()
HD bool ray_tri(f3 v0, f3 v1, f3 v2, f3 o, f3 d, float tmin, float tmax, float& t, float& b1, float& b2)
{
    f3 e1 = v1 - v0, e2 = v2 - v0;
    f3 p = crossf(d, e2);
    float det = dotf(e1, p);
    if (det == 0.0f) return false;
    float inv = 1.0f / det;
    f3 tv = o - v0;
    float u = dotf(tv, p) * inv;
    if (!(u >= 0.0f && u <= 1.0f)) return false;
    f3 q = crossf(tv, e1);
    float v = dotf(d, q) * inv;
    if (!(v >= 0.0f && u + v <= 1.0f)) return false;
    float tt = dotf(e2, q) * inv;
    if (!(tt > tmin && tt < tmax)) return false;
    t = tt; b1 = u; b2 = v;
    return true;
}

// ---- per-lane traversal stack in LDS, element i of this lane at stack[i * BLOCK] (conflict-free columns).  Small scenes that
// are staged into LDS use 16-bit entries (node index < 32768, or leaf: bit15 | count << 13 | first triangle < 8192), which
// halves the stack's LDS footprint; everything else uses the 32-bit child refs as they are.
template <typename ST> HD ST stack_enc(uint32_t ref);
template <> HD uint32_t stack_enc<uint32_t>(uint32_t ref) { return ref; }
template <> HD uint16_t stack_enc<uint16_t>(uint32_t ref)
{
    return (uint16_t)((ref & HJR_LEAF_FLAG) ? (0x8000u | (((ref >> 27) & 3u) << 13) | (ref & 0x1fffu)) : ref);
}
HD uint32_t stack_dec(uint32_t r) { return r; }
HD uint32_t stack_dec(uint16_t r16)
{
    const uint32_t r = r16;
    return (r & 0x8000u) ? (HJR_LEAF_FLAG | (((r >> 13) & 3u) << 27) | (r & 0x1fffu)) : r;
}

// One lane's traversal stack.  SHORT == 0: every entry in LDS (column of this lane).  SHORT > 0 (kernels that read the BVH from
// memory): only the top-of-tree SHORT entries are in LDS, deeper ones overflow into a per-lane column of a global buffer
// ([level][lane], coalesced when neighbouring lanes overflow together).  The exact worst-case depth of a BVH4 over a million
// triangles is ~46 entries, traversal rarely needs more than a dozen: with the whole stack in LDS the stacks, not the
// registers, capped the occupancy at 3 workgroups per CU.
#ifndef HJR_SHORT_STACK
#define HJR_SHORT_STACK 16
#endif
template <typename E, int BLOCK_, int SHORT>
struct LaneStack {
    E* lds;
    uint32_t* spill;
    uint32_t spill_stride;
    HD void put(int i, uint32_t ref)
    {
        if (SHORT == 0 || i < SHORT) lds[i * BLOCK_] = stack_enc<E>(ref);
        else spill[(size_t)(i - SHORT) * spill_stride] = ref;
    }
    HD uint32_t get(int i) const
    {
        if (SHORT == 0 || i < SHORT) return stack_dec(lds[i * BLOCK_]);
        return spill[(size_t)(i - SHORT) * spill_stride];
    }
};

// ---- box-test side of a ray.  The slab test only has to be conservative (boxes are padded): it uses the
// 1-ulp hardware reciprocal and (plane - o) * inv evaluated as fma(plane, inv, -o * inv).  Direction components smaller than
// 1e-30 are clamped (sign kept) so that inv stays finite and no inf - inf can appear for axis-parallel rays.
struct BoxRay {
    f3 inv, oi;
    uint32_t sx, sy, sz; // BVH4 only: 1 when the direction component is negative (near plane row = hi)
};
HD float box_dir(float d) { return (fabsf(d) < 1e-30f) ? copysignf(1e-30f, d) : d; }
HD BoxRay box_ray(f3 o, f3 d)
{
    BoxRay r;
    const f3 dd = V(box_dir(d.x), box_dir(d.y), box_dir(d.z));
    r.inv = V(__builtin_amdgcn_rcpf(dd.x), __builtin_amdgcn_rcpf(dd.y), __builtin_amdgcn_rcpf(dd.z));
    r.oi = V(-o.x * r.inv.x, -o.y * r.inv.y, -o.z * r.inv.z);
    r.sx = dd.x < 0.0f ? 1u : 0u; r.sy = dd.y < 0.0f ? 1u : 0u; r.sz = dd.z < 0.0f ? 1u : 0u; // dead code in the BVH2 kernels
    return r;
}

#define HJR_TRAV_DONE 0xffffffffu
// One inner-node step: tests the children of node `cur` against [tmin, tfar], continues with the nearest hit child, pushes
// the other hit children, or pops (HJR_TRAV_DONE when the stack is empty).  Returns the number of boxes tested.
template <int WIDTH, int BLOCK, typename ST>
HD uint32_t node_step(const float4* nodes, uint32_t& cur, const BoxRay& R, float tmin, float tfar, ST& stack, int& sp)
{
    if constexpr (WIDTH == 2) {
    const float4* nd = nodes + cur * HJR_NODE2_F4;
    const float4 q0 = nd[0], q1 = nd[1], q2 = nd[2], q3 = nd[3];
    const f3 inv = R.inv, oi = R.oi;
    float t0 = fmaf(q0.x, inv.x, oi.x), t1 = fmaf(q0.w, inv.x, oi.x);
    float lo0 = fminf(t0, t1), hi0 = fmaxf(t0, t1);
    t0 = fmaf(q0.y, inv.y, oi.y); t1 = fmaf(q1.x, inv.y, oi.y);
    lo0 = fmaxf(lo0, fminf(t0, t1)); hi0 = fminf(hi0, fmaxf(t0, t1));
    t0 = fmaf(q0.z, inv.z, oi.z); t1 = fmaf(q1.y, inv.z, oi.z);
    lo0 = fmaxf(lo0, fminf(t0, t1)); hi0 = fminf(hi0, fmaxf(t0, t1));
    lo0 = fmaxf(lo0, tmin); hi0 = fminf(hi0, tfar);
    t0 = fmaf(q1.z, inv.x, oi.x); t1 = fmaf(q2.y, inv.x, oi.x);
    float lo1 = fminf(t0, t1), hi1 = fmaxf(t0, t1);
    t0 = fmaf(q1.w, inv.y, oi.y); t1 = fmaf(q2.z, inv.y, oi.y);
    lo1 = fmaxf(lo1, fminf(t0, t1)); hi1 = fminf(hi1, fmaxf(t0, t1));
    t0 = fmaf(q2.x, inv.z, oi.z); t1 = fmaf(q2.w, inv.z, oi.z);
    lo1 = fmaxf(lo1, fminf(t0, t1)); hi1 = fminf(hi1, fmaxf(t0, t1));
    lo1 = fmaxf(lo1, tmin); hi1 = fminf(hi1, tfar);
    const bool h0 = lo0 <= hi0, h1 = lo1 <= hi1; // conservative through the 2^-15 box padding (>= 16x the rounding error of t)
    const uint32_t c0 = f2bits(q3.x), c1 = f2bits(q3.y);
    if (h0 && h1) {
        const bool swap = lo1 < lo0;
        stack.put(sp, swap ? c0 : c1);
        sp++;
        cur = swap ? c1 : c0;
    } else if (h0) cur = c0;
    else if (h1) cur = c1;
    else if (sp > 0) { sp--; cur = stack.get(sp); }
    else cur = HJR_TRAV_DONE;
    return 2u;
    } else {
    const float4* nd = nodes + cur * HJR_NODE4_F4;
    const f3 inv = R.inv, oi = R.oi;
    const float INF = bits2f(0x7f800000u);
    // near / far plane rows picked by the ray's direction signs: no min/max per axis
    const float4 nx = nd[0 + R.sx], fx = nd[1 - R.sx];
    const float4 ny = nd[2 + R.sy], fy = nd[3 - R.sy];
    const float4 nz = nd[4 + R.sz], fz = nd[5 - R.sz];
    const float4 rr = nd[6];
#define HJR_CHILD(c, C)                                                                                                  \
    float tn##C = fmaxf(fmaxf(fmaf(nx.c, inv.x, oi.x), fmaf(ny.c, inv.y, oi.y)), fmaxf(fmaf(nz.c, inv.z, oi.z), tmin)); \
    const float tf##C = fminf(fminf(fmaf(fx.c, inv.x, oi.x), fmaf(fy.c, inv.y, oi.y)), fminf(fmaf(fz.c, inv.z, oi.z), tfar)); \
    const bool h##C = tn##C <= tf##C;                                                                                   \
    tn##C = h##C ? tn##C : INF;
    HJR_CHILD(x, 0) HJR_CHILD(y, 1) HJR_CHILD(z, 2) HJR_CHILD(w, 3)
#undef HJR_CHILD
    const uint32_t r0 = f2bits(rr.x), r1 = f2bits(rr.y), r2 = f2bits(rr.z), r3 = f2bits(rr.w);
    const float m = fminf(fminf(tn0, tn1), fminf(tn2, tn3));
    // nearest hit child first (ties: lowest slot); the other hit children are pushed in slot order
    const int sel = (tn0 == m) ? 0 : ((tn1 == m) ? 1 : ((tn2 == m) ? 2 : 3));
    if (h3 && sel != 3) { stack.put(sp, r3); sp++; }
    if (h2 && sel != 2) { stack.put(sp, r2); sp++; }
    if (h1 && sel != 1) { stack.put(sp, r1); sp++; }
    if (h0 && sel != 0) { stack.put(sp, r0); sp++; }
    if (h0 || h1 || h2 || h3) cur = (sel == 0) ? r0 : ((sel == 1) ? r1 : ((sel == 2) ? r2 : r3));
    else if (sp > 0) { sp--; cur = stack.get(sp); }
    else cur = HJR_TRAV_DONE;
    return 4u;
    }
}

struct Hit { float t, b1, b2; uint32_t k, prim; };

// stack: this lane's column of the LDS stack, element i at stack[i * BLOCK]
template <bool ANY, bool STATS, int WIDTH, int BLOCK, typename ST>
HD bool traverse(const float4* nodes, const float4* tris, f3 o, f3 d, float tmin, float tmax, Hit& hit, ST& stack, Counters& cnt)
{
    const BoxRay R = box_ray(o, d);
    int sp = 0;
    uint32_t cur = 0;
    hit.prim = 0xffffffffu;
    hit.t = tmax;
    for (;;) {
        while (!(cur & HJR_LEAF_FLAG)) { // descend through inner nodes until this lane holds a leaf (or is done)
            const uint32_t nb = node_step<WIDTH, BLOCK, ST>(nodes, cur, R, tmin, hit.t, stack, sp);
            if (STATS) cnt.box += nb;
        }
        if (cur == HJR_TRAV_DONE) break;
        const uint32_t first = cur & 0x07ffffffu, count = (cur >> 27) & 15u;
        for (uint32_t i = 0; i < count; i++) {
            const float4* g = tris + (first + i) * HJR_TRI_F4;
            const float4 g0 = g[0], g1 = g[1], g2 = g[2];
            float t, b1, b2;
            if (STATS) cnt.tri++;
            if (ray_tri(V(g0.x, g0.y, g0.z), V(g0.w, g1.x, g1.y), V(g1.z, g1.w, g2.x), o, d, tmin, tmax, t, b1, b2)) {
                if (ANY) return true;
                const uint32_t prim = f2bits(g2.y);
                // order-independent closest-hit rule: smaller t wins; equal t -> smaller global prim id
                if (hit.prim == 0xffffffffu || t < hit.t || (t == hit.t && prim < hit.prim)) {
                    hit.t = t; hit.b1 = b1; hit.b2 = b2; hit.k = first + i; hit.prim = prim;
                }
            }
        }
        if (sp == 0) break;
        sp--;
        cur = stack.get(sp);
    }
    return hit.prim != 0xffffffffu;
}

// Fused traversal of two rays per lane in ONE loop: ray A = the pending NEE shadow ray of the bounce just shaded (any-hit),
// ray B = the next closest-hit ray (continuation or a regenerated primary ray).  A lane moves on to B the moment its A is
// resolved, without waiting for the rest of the wave, so the wave's trip count is max_lanes(tripsA + tripsB) instead of
// max(tripsA) + max(tripsB) — the SIMT cost of per-lane trip-count variance drops by ~1/3 (profiles/r01_experiments.md).
// Results are identical to two separate traversals.
//
// Straggler carry-over (CARRY > 0): the loop also ends when at most CARRY lanes are still traversing (and at least one
// lane of this round has finished).  Those lanes keep their traversal state (TravCarry + hit + their LDS stack column),
// skip the shading that follows and resume in the next round next to the other lanes' new rays: the wave's trip count per
// round is set by the (64 - CARRY)-th slowest lane instead of the slowest one.  Per-lane results do not change.  The
// threshold trades traversal lane-occupancy against shading lane-occupancy (profiles/r01_experiments.md): 8 for the
// LDS-resident scenes (shading-heavy), 32 when nodes come from memory (traversal-heavy).
#ifndef HJR_CARRY_LDS
#define HJR_CARRY_LDS 8
#endif
#ifndef HJR_CARRY_MEM
#define HJR_CARRY_MEM 32
#endif
struct TravCarry { uint32_t cur; int sp, phase; };
template <bool STATS, int WIDTH, int BLOCK, typename ST, int CARRY>
HD bool traverse_fused(const float4* nodes, const float4* tris, const bool a_valid, const f3 ao, const f3 ad, const float a_tmax, const bool b_valid,
                       const f3 bo, const f3 bd, bool& occluded, Hit& hit, ST& stack, Counters& ca, Counters& cb, const bool resume, TravCarry& tc)
{
    const float tmin = 0.001f;
    int phase, sp;
    uint32_t cur;
    if (CARRY > 0 && resume) { phase = tc.phase; sp = tc.sp; cur = tc.cur; } // occluded / hit are the caller's, kept across rounds
    else {
        occluded = false;
        hit.prim = 0xffffffffu;
        hit.t = 1e16f;
        phase = a_valid ? 0 : (b_valid ? 1 : 2);
        sp = 0;
        cur = (phase < 2) ? 0u : HJR_TRAV_DONE;
    }
    f3 o = (phase == 0) ? ao : bo;
    f3 d = (phase == 0) ? ad : bd;
    BoxRay R = box_ray(o, d);
    const int n_start = CARRY > 0 ? __popcll(__ballot(phase < 2)) : 0;
#ifdef HJR_TIMING
    unsigned long long t_node = 0, t_leaf = 0, t_last = __builtin_amdgcn_s_memtime();
#endif
    for (;;) {
        if (CARRY > 0) {
            const int n_act = __popcll(__ballot(phase < 2));
            if (n_act == 0 || (n_act <= CARRY && n_act < n_start)) break;
        } else if (__ballot(phase < 2) == 0ull) break;
        if (phase < 2) {
        // "while-while" traversal: every lane first descends through inner nodes until it holds a leaf (or is out of work) ...
        while (!(cur & HJR_LEAF_FLAG)) {
            const float tfar = (phase == 0) ? a_tmax : hit.t;
            const uint32_t nb = node_step<WIDTH, BLOCK, ST>(nodes, cur, R, tmin, tfar, stack, sp);
            if (STATS) { if (phase == 0) ca.box += nb; else cb.box += nb; }
        }
#ifdef HJR_TIMING
        { __builtin_amdgcn_sched_barrier(0); unsigned long long now_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_sched_barrier(0); t_node += now_ - t_last; t_last = now_; }
#endif
        // ... then all lanes test their leaf's triangles together
        bool done = (cur == HJR_TRAV_DONE);
        if (!done) {
            const uint32_t first = cur & 0x07ffffffu, count = (cur >> 27) & 15u;
            const float tri_tmax = (phase == 0) ? a_tmax : 1e16f;
            for (uint32_t i = 0; i < count; i++) {
                const float4* g = tris + (first + i) * HJR_TRI_F4;
                const float4 g0 = g[0], g1 = g[1], g2 = g[2];
                float t, b1, b2;
                if (STATS) { if (phase == 0) ca.tri++; else cb.tri++; }
                if (ray_tri(V(g0.x, g0.y, g0.z), V(g0.w, g1.x, g1.y), V(g1.z, g1.w, g2.x), o, d, tmin, tri_tmax, t, b1, b2)) {
                    if (phase == 0) { occluded = true; done = true; break; }
                    const uint32_t prim = f2bits(g2.y);
                    // order-independent closest-hit rule: smaller t wins; equal t -> smaller global prim id
                    if (hit.prim == 0xffffffffu || t < hit.t || (t == hit.t && prim < hit.prim)) {
                        hit.t = t; hit.b1 = b1; hit.b2 = b2; hit.k = first + i; hit.prim = prim;
                    }
                }
            }
            if (!done) {
                if (sp > 0) { sp--; cur = stack.get(sp); }
                else done = true;
            }
        }
        if (done) {
            if (phase == 0 && b_valid) { // this lane's shadow ray is resolved: start its closest-hit ray right away
                phase = 1;
                o = bo; d = bd;
                R = box_ray(o, d);
                sp = 0; cur = 0;
            } else { phase = 2; cur = HJR_TRAV_DONE; }
        }
#ifdef HJR_TIMING
        { __builtin_amdgcn_sched_barrier(0); unsigned long long now_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_sched_barrier(0); t_leaf += now_ - t_last; t_last = now_; }
#endif
    } }
#ifdef HJR_TIMING
    ca.t_node = t_node; ca.t_leaf = t_leaf;
#endif
    if (CARRY > 0) { tc.phase = phase; tc.sp = sp; tc.cur = cur; return phase < 2; }
    return false;
}

// ------------------------------------------------------------------ closest-hit / miss programs (build-defined)
struct HitInfo { // the Payload fields the integrators read (kernel/Payload.h:12-42)
    bool is_hit, is_light;
    f3 position, normal, emission;
    Surface surf;
    uint32_t prim;
};

// __closesthit__ch / __miss__ms for a finished closest-hit traversal
template <bool STATS, bool FULL>
HD void hit_program(const KParams& P, const float4* tris, const float4* mats, const Hit& h, const f3 rd, HitInfo& prd, unsigned long long* lc)
{
    if (h.prim == 0xffffffffu) { // __miss__ms: constant sky (use_IBL = false: 1x1 texel scene_sky_default, renderer.h:802-851) * ibl_intensity
        prd.is_hit = false; prd.is_light = false;
        if (FULL && P.sky_tex) prd.emission = sky_fetch(P, rd) * P.ibl_intensity; // tex2D(ibl_texture, u, v) * ibl_intensity
        else prd.emission = V(P.sky[0], P.sky[1], P.sky[2]);
        prd.position = V1(0.0f); prd.normal = V1(0.0f);
        prd.surf.basecolor = V1(0.0f); // Payload default (Payload.h:25)
        prd.surf.metallic = 0.0f; prd.surf.roughness = 0.0f; prd.surf.sheen = 0.0f; prd.surf.clearcoat = 0.0f; prd.surf.ior = 1.0f;
        prd.surf.is_specular = false; prd.surf.is_thinfilm = false;
        prd.prim = 0xffffffffu;
        return;
    }
    // __closesthit__ch: barycentric interpolation of the pre-transformed vertices / normals with (1-b1-b2, b1, b2);
    // the interpolated normal is neither re-normalised nor flipped (stale ptx:1244-1283)
    const float4* g = tris + h.k * HJR_TRI_F4;
    const float4 g0 = g[0], g1 = g[1], g2 = g[2];
    const float4* s = P.tri_shade + (size_t)h.prim * HJR_SHADE_F4;
    const float4 s0 = s[0], s1 = s[1], s2 = s[2], s3 = s[3];
    const float w0 = 1.0f - h.b1 - h.b2;
    prd.is_hit = true;
    prd.position = V(g0.x, g0.y, g0.z) * w0 + V(g0.w, g1.x, g1.y) * h.b1 + V(g1.z, g1.w, g2.x) * h.b2;
    prd.normal = V(s0.x, s0.y, s0.z) * w0 + V(s1.x, s1.y, s1.z) * h.b1 + V(s2.x, s2.y, s2.z) * h.b2;
    const float4* m = mats + f2bits(g2.z) * HJR_MAT_F4; // g2.z == s3.w (material id): this fetch does not wait for the shading record
    const float4 m0 = m[0], m1 = m[1], m2 = m[2], m3 = m[3];
    prd.surf.basecolor = V(m0.x, m0.y, m0.z);
    prd.surf.metallic = m0.w;
    prd.surf.roughness = m1.x; prd.surf.sheen = m1.y; prd.surf.clearcoat = m1.z; prd.surf.ior = m1.w;
    if (FULL && P.tex_desc) { // material textures: factor x texel (glTF 2.0 semantics; build-defined)
        const int bc_tex = (int)f2bits(m3.w), mr_tex = (int)f2bits(m[4].x);
        if (bc_tex >= 0 || mr_tex >= 0) {
            const float tu = s0.w * w0 + s2.w * h.b1 + s3.y * h.b2; // uv0 = (s0.w, s1.w), uv1 = (s2.w, s3.x), uv2 = (s3.y, s3.z)
            const float tv = s1.w * w0 + s3.x * h.b1 + s3.z * h.b2;
            if (bc_tex >= 0) prd.surf.basecolor = prd.surf.basecolor * tex_fetch(P, bc_tex, tu, tv);
            if (mr_tex >= 0) {
                const f3 e = tex_fetch(P, mr_tex, tu, tv);
                prd.surf.roughness = prd.surf.roughness * e.y;
                prd.surf.metallic = prd.surf.metallic * e.z;
            }
        }
    }
    prd.emission = V(m2.y, m2.z, m2.w);
    prd.is_light = f2bits(m3.x) != 0;
    prd.surf.is_specular = f2bits(m3.y) != 0;
    prd.surf.is_thinfilm = f2bits(m3.z) != 0;
    prd.prim = h.prim;
    if (STATS) lc[7] += 1;
}

// RayTrace (rt.h:43-69): stand-alone closest-hit query (used by MIS' BSDF-sampled light ray)
template <bool STATS, bool FULL, int WIDTH, int BLOCK, typename ST>
HD void ray_trace(const KParams& P, const float4* nodes, const float4* tris, const float4* mats, f3 o, f3 d, HitInfo& prd, ST& stack, unsigned long long* lc)
{
    Hit h;
    Counters c; c.box = 0; c.tri = 0;
    traverse<false, STATS, WIDTH, BLOCK, ST>(nodes, tris, o, d, 0.001f, 1e16f, h, stack, c);
    if (STATS) { lc[1] += 1; lc[3] += c.box; lc[4] += c.tri; }
    hit_program<STATS, FULL>(P, tris, mats, h, d, prd, lc);
}

// light_sample (kernel/light_sample.h:9-75) on the per-frame light table
HD f3 light_sample(const KParams& P, const float4* lights, CMJState& st, float& pdf, f3& normal, f3& emission)
{
    float p = cmj_1d(st);
    int index = (int)(p * P.n_lights);
    if (index == (int)P.n_lights) index--;
    const float4* L = lights + (size_t)index * HJR_LIGHT_F4;
    const float4 l0 = L[0], l1 = L[1], l2 = L[2], l3 = L[3], l4 = L[4], l5 = L[5];
    f2 xi = cmj_2d(st);
    float f1 = 1.0f - sqrtf(xi.x);
    float f2_ = sqrtf(xi.x) * (1.0f - xi.y);
    float f3_ = sqrtf(xi.x) * xi.y;
    const f3 light_position = V(l0.x, l0.y, l0.z) * f1 + V(l1.x, l1.y, l1.z) * f2_ + V(l2.x, l2.y, l2.z) * f3_;
    normal = normalize(V(l3.x, l3.y, l3.z) * f1 + V(l4.x, l4.y, l4.z) * f2_ + V(l5.x, l5.y, l5.z) * f3_);
    pdf = l0.w;
    emission = V(l1.w, l2.w, l3.w);
    return light_position;
}

// ------------------------------------------------------------------ the megakernel
#define HJR_BLOCK 256
#define HJR_INTEGRATOR_NEE_ 0
#define HJR_INTEGRATOR_PT_ 1
#define HJR_INTEGRATOR_MIS_ 2

struct PathState {
    f3 ro, rd, thr, L;
    uint32_t rng_depth; // CMJState.depth; the other CMJState fields are functions of (frame, spp, s, seed, pixel): rebuilt on use
    int depth;
};

// CMJState of the path that is running sample s of pixel (px, py) (build-defined seeding)
HD CMJState path_rng(const KParams& P, uint32_t px, uint32_t py, uint32_t s, uint32_t depth)
{
    CMJState st;
    st.n_spp = (unsigned long long)P.frame * (unsigned long long)P.spp + (unsigned long long)s;
    st.scramble = P.seed;
    st.depth = depth;
    st.image_idx = px + py * P.width;
    return st;
}

// __raygen__rg for one sample (build-defined; SURVEY §8a a1/a2, stale ptx:33-106): CMJ stream keyed by
// (frame * spp + s, seed, pixel), first 2-D draw = sub-pixel jitter, u = (2(x+jx) - W) / H, v = (2(y+jy) - H) / H
HD void start_path(const KParams& P, PathState& ps, uint32_t px, uint32_t py, uint32_t s)
{
    CMJState st = path_rng(P, px, py, s, 0u);
    f2 j = cmj_2d(st);
    ps.rng_depth = st.depth;
    float W = (float)P.width, H = (float)P.height;
    float u = (2.0f * ((float)px + j.x) - W) / H;
    float v = (2.0f * ((float)py + j.y) - H) / H;
    f3 cd = V(P.cam_dir[0], P.cam_dir[1], P.cam_dir[2]);
    f3 cu = V(P.cam_up[0], P.cam_up[1], P.cam_up[2]);
    f3 cr = V(P.cam_right[0], P.cam_right[1], P.cam_right[2]);
    // the ray origin is the (wave-uniform) camera position: not stored per lane, see `fresh` in the kernel
    ps.rd = normalize(cd * P.cam_f + cr * u + cu * v);
    ps.thr = V1(1.0f);
    ps.depth = 0;
}

#ifndef HJR_MIN_WAVES
#define HJR_MIN_WAVES 4 /* memory-path kernels: 128 VGPRs = 4 waves per SIMD (with the short LDS stacks the registers, not LDS, set the occupancy; 3 waves: 329 ms, 4: 279, 5: 283, 6: 315 on the 1 M-triangle scene) */
#endif
// Dynamic LDS: [traversal stacks: stack_depth x BLOCK uint32][nodes][tri_geom]  (the last two only when LDSBVH).
// LDSBVH: the whole BVH + leaf-order triangles are staged into LDS once per persistent workgroup (coalesced dwordx4 loads),
// so every traversal step reads LDS (ds_read_b128, ~100-cycle latency) instead of L2 (~500+ cycles under load).  One
// workgroup of BLOCK threads per CU shares the copy.  Chosen by the host when the scene fits (hjr_device.hip).
extern __shared__ float4 hjr_smem[];

// AOVS = the "full" variant: albedo / normal AOV sums, material textures and the equirect sky texture; the lean variant
// (colour only, untextured scene, constant sky) saves registers and is what the headline benchmark runs
template <int INTEGRATOR, bool STATS, int BLOCK, bool LDSBVH, bool STACK16, int WIDTH, bool AOVS>
__global__ void __launch_bounds__(BLOCK, (LDSBVH ? 1 : HJR_MIN_WAVES)) hjr_render_kernel(const KParams P)
{
    typedef typename std::conditional<STACK16, uint16_t, uint32_t>::type SE; // stack entry type
    constexpr int SHORT = LDSBVH ? 0 : HJR_SHORT_STACK;
    typedef LaneStack<SE, BLOCK, SHORT> ST;
    ST stack;
    stack.lds = reinterpret_cast<SE*>(hjr_smem) + threadIdx.x;
    stack.spill = P.stack_spill + (blockIdx.x * BLOCK + threadIdx.x);
    stack.spill_stride = P.spill_stride;
    const uint32_t lane = threadIdx.x & 63u;
    const float4* nodes = P.nodes;
    const float4* tris = P.tri_geom;
    const float4* mats = P.materials;
    const float4* lights = P.lights;
    if (LDSBVH) {
        float4* l_nodes = hjr_smem + (BLOCK * P.stack_depth * (uint32_t)sizeof(SE) + 15u) / 16u;
        float4* l_tris = l_nodes + P.n_node_f4;
        for (uint32_t i = threadIdx.x; i < P.n_node_f4; i += BLOCK) l_nodes[i] = P.nodes[i];
        for (uint32_t i = threadIdx.x; i < P.n_tri_f4; i += BLOCK) l_tris[i] = P.tri_geom[i];
        // the (small) material and light tables ride along: one LDS read instead of an L2 round trip per shaded hit
        float4* l_mats = l_tris + P.n_tri_f4;
        float4* l_lights = l_mats + P.n_mat_f4;
        for (uint32_t i = threadIdx.x; i < P.n_mat_f4; i += BLOCK) l_mats[i] = P.materials[i];
        for (uint32_t i = threadIdx.x; i < P.n_light_f4; i += BLOCK) l_lights[i] = P.lights[i];
        __syncthreads();
        nodes = l_nodes;
        tris = l_tris;
        mats = l_mats;
        lights = l_lights;
    }

    unsigned long long lc[HJR_NSTAT];
    if (STATS) for (int i = 0; i < HJR_NSTAT; i++) lc[i] = 0;

    bool has_item = false, dead = false, path_live = false;
    bool fin_pending = false;   // a finished path whose L still waits for its last NEE shadow ray
    bool write_pending = false; // the item's last sample is finished; sums go out once fin_pending is resolved
    bool sh_valid = false;      // pending NEE shadow ray of the bounce shaded in the previous iteration
    // Register diet (the LDS variant runs at 128 VGPRs): pixel and chunk share one word; ps.ro doubles as the origin of the
    // pending shadow ray (a regenerated path starts at the wave-uniform camera position: `fresh`); ps.L keeps the finished
    // path's radiance while fin_pending (the new path's L is 0 until that is resolved).
    bool fresh = true;          // the closest-hit ray of this lane starts at the camera
    bool inflight = false;      // carry-over: this lane's traversal continues in the next round (it skips everything else)
    bool tracing = false, occluded = false;
    Hit h;
    TravCarry tc; tc.cur = HJR_TRAV_DONE; tc.sp = 0; tc.phase = 2;
    uint32_t item = 0;          // px | py << 13 | chunk << 26
    uint32_t s = 0;
    uint32_t w_next = 0, w_end = 0; // this wave's private item range (wave-uniform)
    uint32_t it_cost = 0;           // closest-hit rays traced for the current item
    f3 sumL = V1(0.0f), sumA = V1(0.0f), sumN = V1(0.0f);
    f3 sh_d = V1(0.0f), sh_contrib = V1(0.0f);
    float sh_tmax = 0.0f;
#define HJR_PX (item & 0x1fffu)
#define HJR_PY ((item >> 13) & 0x1fffu)
#define HJR_CHUNK (item >> 26)
#define HJR_S_END min((HJR_CHUNK + 1u) * P.chunk_spp, P.spp)
    PathState ps;
    ps.ro = ps.rd = ps.thr = ps.L = V1(0.0f);
    ps.depth = 0;
    ps.rng_depth = 0;
    const float inv_spp = 1.0f / (float)P.spp;
#ifdef HJR_TIMING
    // diagnostic build: wave-clock shares of the loop's phases, summed per wave into P.stats[10..15] (never in the shipped build)
    unsigned long long tk[6] = { 0, 0, 0, 0, 0, 0 }, tk6 = 0, tk7 = 0, tx[4] = { 0, 0, 0, 0 };
#define HJR_TICKX(i) { __builtin_amdgcn_sched_barrier(0); unsigned long long now_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_sched_barrier(0); tx[i] += now_ - tstamp; tstamp = now_; }
    bool dg_shade = false, dg_ms = false, dg_glass = false;
    unsigned long long oc[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }; // wave rounds, lanes tracing closest, lanes with a shadow ray, lanes shading, msGGX lanes, glass lanes, rounds with shading, lanes serviced
    unsigned long long tstamp = __builtin_amdgcn_s_memtime();
#define HJR_TICK(i) { __builtin_amdgcn_sched_barrier(0); unsigned long long now_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_sched_barrier(0); tk[i] += now_ - tstamp; tstamp = now_; }
#elif defined(HJR_MARK)
// static-analysis build: region markers in the ISA listing (count instructions between them), never shipped
#define HJR_TICK(i) { __builtin_amdgcn_sched_barrier(0); asm volatile("; HJRMARK T" #i); __builtin_amdgcn_sched_barrier(0); }
#define HJR_TICKX(i) { __builtin_amdgcn_sched_barrier(0); asm volatile("; HJRMARK X" #i); __builtin_amdgcn_sched_barrier(0); }
#else
#define HJR_TICK(i)
#define HJR_TICKX(i)
#endif

    for (;;) {
#ifdef HJR_TIMING
        { // wave-uniform occupancy sums of the previous round (all lanes are here; flags are lane-private)
            const int n_sh = __popcll(__ballot(dg_shade));
            oc[3] += n_sh; oc[4] += __popcll(__ballot(dg_ms)); oc[5] += __popcll(__ballot(dg_glass)); oc[6] += n_sh > 0 ? 1 : 0;
            dg_shade = dg_ms = dg_glass = false;
        }
#endif
        HJR_TICKX(0)
        // NaN/Inf guard + ordered accumulation of one finished sample
        auto finish_sample = [&](f3 L) {
            float sum = L.x + L.y + L.z;
            if (!(sum - sum == 0.0f)) { L = V1(0.0f); if (STATS) lc[9] += 1; }
            sumL = sumL + L;
            if (STATS) lc[0] += 1;
        };
        // sample bookkeeping at the end of a path (independent of the radiance value)
        auto close_sample = [&]() {
            s++;
            path_live = false;
            if (s == HJR_S_END) { write_pending = true; has_item = false; }
        };
        auto write_out = [&]() {
            const size_t pix = (size_t)HJR_PX + (size_t)HJR_PY * P.width;
            if (P.n_chunks == 1u) { // the item is the whole pixel: mean = chunk sum * (1 / spp)
                P.aov_color[pix] = make_float4(sumL.x * inv_spp, sumL.y * inv_spp, sumL.z * inv_spp, 1.0f);
                if (AOVS && P.aov_albedo) P.aov_albedo[pix] = make_float4(sumA.x * inv_spp, sumA.y * inv_spp, sumA.z * inv_spp, 1.0f);
                if (AOVS && P.aov_normal) P.aov_normal[pix] = make_float4(sumN.x * inv_spp, sumN.y * inv_spp, sumN.z * inv_spp, 1.0f);
            } else { // chunk sum -> HBM; hjr_finalize_kernel adds the chunks of a pixel in chunk order
                const size_t slot = (size_t)HJR_CHUNK * ((size_t)P.width * P.height) + pix;
                P.part_color[slot] = make_float4(sumL.x, sumL.y, sumL.z, 0.0f);
                if (AOVS && P.part_albedo) P.part_albedo[slot] = make_float4(sumA.x, sumA.y, sumA.z, 0.0f);
                if (AOVS && P.part_normal) P.part_normal[slot] = make_float4(sumN.x, sumN.y, sumN.z, 0.0f);
            }
            write_pending = false;
        };

        // ---- Russian roulette (rt.h:173-179) with in-place path regeneration: a lane whose path dies here starts its
        //      next sample immediately, so it still has a closest-hit ray for this iteration's trace.  The dead path's
        //      radiance is final only after its pending shadow ray (fused into the same trace) is resolved.
        if (!inflight) {
            if (has_item && path_live) { // continuing path
                const float russian_p = fmaxf(ps.thr.x, fmaxf(ps.thr.y, ps.thr.z));
                CMJState rr = path_rng(P, HJR_PX, HJR_PY, s, ps.rng_depth);
                const float xi_rr = cmj_1d(rr);
                ps.rng_depth = rr.depth;
                if (russian_p < xi_rr) {
                    if (sh_valid) { fin_pending = true; close_sample(); } // radiance final once the pending shadow ray is resolved
                    else { // nothing pending: the sample is final now, and if it was the item's last one the lane refills below
                        finish_sample(ps.L);
                        ps.L = V1(0.0f);
                        close_sample();
                        if (write_pending) write_out();
                    }
                } else ps.thr = ps.thr / russian_p;
            }
        }

        // ---- ray-queue refill (ballot + mbcnt prefix): idle lanes take consecutive items from the wave's private range
        //      [w_next, w_end); when it runs dry the wave fetches the next 64 items with ONE atomic on the global head.
        {
            const bool need = !has_item && !dead && !write_pending && !fin_pending && !sh_valid;
            const unsigned long long m = __ballot(need);
            if (m) {
                const uint32_t n = (uint32_t)__popcll(m);
                const uint32_t prefix = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                uint32_t q = w_next + prefix;          // wave-uniform w_next / w_end
                const uint32_t have = w_end - w_next;  // items left in the private range
                if (n > have) {                        // not enough: lanes beyond `have` come from a fresh range
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(P.queue_head, 64u);
                    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                    if (prefix >= have) q = base + (prefix - have);
                    w_next = base + (n - have);
                    w_end = base + 64u;
                } else w_next += n;
                // measured cost of a tile (orders the tiles of the next frame, hjr_cost_hist_kernel): a lane sums the rays of its
                // consecutive items of one tile and flushes when it moves on; lanes leaving the same tile together (the usual
                // case) share one atomic.  All lanes are here (m is wave-uniform), so the shuffles below are well defined.
                const uint32_t old_tile = (HJR_PY / HJR_TILE) * P.tiles_x + HJR_PX / HJR_TILE;
                uint32_t new_tile = 0xffffffffu;
                if (need && q < P.n_owned_items) new_tile = P.tile_order ? P.tile_order[(q >> 6) / P.n_chunks] : ((q >> 6) / P.n_chunks) * P.world + P.rank;
                bool flush = need && P.tile_cost && it_cost != 0u && new_tile != old_tile;
                while (__ballot(flush)) {
                    const int leader = __ffsll((long long)__ballot(flush)) - 1;
                    const uint32_t t = (uint32_t)__shfl((int)old_tile, leader);
                    const bool mine = flush && old_tile == t;
                    uint32_t v = mine ? it_cost : 0u;
                    for (int off = 32; off > 0; off >>= 1) v += (uint32_t)__shfl_xor((int)v, off);
                    if ((int)lane == leader) atomicAdd(&P.tile_cost[t / P.world], v);
                    if (mine) { it_cost = 0u; flush = false; }
                }
                if (need) {
                    if (q < P.n_owned_items) {
                        // item q = ((owned tile * n_chunks) + chunk) * 64 + pixel-in-tile: the 64 lanes of a wave start on one
                        // tile and one sample chunk (coherent primary rays)
                        const uint32_t tc = q >> 6;
                        const uint32_t tile = new_tile;
                        const uint32_t chunk = tc % P.n_chunks;
                        const uint32_t tx = tile % P.tiles_x, ty = tile / P.tiles_x;
                        const uint32_t px = tx * HJR_TILE + (q & 7u);
                        const uint32_t py = ty * HJR_TILE + ((q >> 3) & 7u);
                        if (px < P.width && py < P.height) {
                            has_item = true; path_live = false;
                            item = px | (py << 13) | (chunk << 26);
                            s = chunk * P.chunk_spp;
                            sumL = V1(0.0f); sumA = V1(0.0f); sumN = V1(0.0f);
                        }
                    } else dead = true;
                }
            }
            if (__ballot(!dead) == 0ull) break; // a lane only dies with nothing pending
        }

        if (!inflight) {
            if (has_item && !path_live) {
                start_path(P, ps, HJR_PX, HJR_PY, s);
                if (!fin_pending) ps.L = V1(0.0f); // while fin_pending, ps.L still belongs to the finished path
                path_live = true; fresh = true;
                // the new path's own roulette draw: throughput is (1,1,1), so russian_p = 1 > xi for every xi in [0,1) and
                // thr / 1 == thr; only the stream position moves
                ps.rng_depth += 1u;
            }
            tracing = has_item;
        }

        HJR_TICK(0)
#ifdef HJR_TIMING
        oc[0] += 1; oc[1] += __popcll(__ballot(tracing)); oc[2] += __popcll(__ballot(sh_valid));
#endif
        // ---- one fused traversal: pending shadow ray (TraceOcculution, rt.h:236-243) then closest-hit ray (RayTrace, rt.h:182-189)
        {
            Counters ca, cb; ca.box = ca.tri = cb.box = cb.tri = 0;
            const f3 cam_o = V(P.cam_pos[0], P.cam_pos[1], P.cam_pos[2]);
            inflight = traverse_fused<STATS, WIDTH, BLOCK, ST, (LDSBVH ? HJR_CARRY_LDS : HJR_CARRY_MEM)>(nodes, tris, sh_valid, ps.ro, sh_d, sh_tmax, tracing, fresh ? cam_o : ps.ro, ps.rd, occluded, h, stack, ca, cb, inflight, tc);
#ifdef HJR_TIMING
            tk6 += ca.t_node; tk7 += ca.t_leaf;
#endif
            if (STATS) { // tests are counted round by round, rays when they are resolved
                lc[5] += ca.box; lc[6] += ca.tri; lc[3] += cb.box; lc[4] += cb.tri;
                if (!inflight) { if (sh_valid) lc[2] += 1; if (tracing) lc[1] += 1; }
            }
        }
        HJR_TICK(1)
#ifdef HJR_TIMING
        oc[7] += __popcll(__ballot(!inflight && (tracing || sh_valid)));
#endif
        if (inflight) continue;
        if (sh_valid) { // `if (!light_shot.is_hit) LTE += ...` (rt.h:245-259), added to the path the shadow ray belongs to
            if (!occluded) ps.L = ps.L + sh_contrib; // ps.L is the finished path's radiance while fin_pending
            sh_valid = false;
        }
        if (fin_pending) {
            finish_sample(ps.L);
            ps.L = V1(0.0f); // from here on ps.L belongs to the path that was regenerated (or to nothing)
            fin_pending = false;
            if (write_pending) write_out();
        }
        HJR_TICKX(1)

        if (tracing) {
            it_cost++;
            HitInfo prd;
            hit_program<STATS, AOVS>(P, tris, mats, h, ps.rd, prd, lc);
            HJR_TICKX(2)
            if (AOVS && ps.depth == 0) { sumA = sumA + prd.surf.basecolor; sumN = sumN + prd.normal; } // rt.h:191-194
            if (!prd.is_hit || prd.is_light) {
                // NEE / MIS count emission only at depth 0 (rt.h:196-208, 318-330); Pathtrace always (rt.h:118-126)
                if (INTEGRATOR == HJR_INTEGRATOR_PT_ || ps.depth == 0) ps.L = ps.L + ps.thr * prd.emission;
                finish_sample(ps.L);
                close_sample();
                if (write_pending) write_out();
            } else {
                HJR_TICK(2)
#ifdef HJR_TIMING
                dg_shade = true; dg_ms = !prd.surf.is_specular && prd.surf.metallic > 0.5f; dg_glass = prd.surf.is_specular;
#endif
                CMJState st = path_rng(P, HJR_PX, HJR_PY, s, ps.rng_depth);
                const Surface& sf = prd.surf;
                f3 t, b;
                const f3 n = prd.normal;
                orthonormal_basis(n, t, b);
                const f3 local_wo = world_to_local(-ps.rd, t, n, b);

                if (INTEGRATOR != HJR_INTEGRATOR_PT_ && P.n_lights >= 1u) { // light_prim_count < 1: no contribution (UB in the reference)
                    float light_pdf;
                    f3 light_color, light_normal;
                    const f3 light_position = light_sample(P, lights, st, light_pdf, light_normal, light_color);
                    if (STATS) lc[8] += 1;
                    const f3 so = prd.position;
                    f3 sd;
                    float light_distance;
                    if (INTEGRATOR == HJR_INTEGRATOR_NEE_) { // rt.h:230-233
                        sd = normalize(light_position - so);
                        light_distance = length3(light_position - so);
                    } else { // rt.h:352-354
                        sd = light_position - so;
                        light_distance = length3(sd);
                        sd = normalize(sd);
                    }
                    // the contribution is fully determined here; only whether it is added depends on the shadow ray, which is
                    // traced fused with the next closest-hit ray at the top of the next iteration
                    const float cosine1 = absdot(n, sd);
                    const float cosine2 = absdot(light_normal, -sd);
                    const f3 local_wi = world_to_local(sd, t, n, b);
                    const f3 bsdf = bsdf_eval(P, sf, local_wo, local_wi);
                    const float G = cosine2 / (light_distance * light_distance);
                    if (INTEGRATOR == HJR_INTEGRATOR_NEE_) {
                        sh_contrib = (ps.thr * ((bsdf * G * cosine1) / light_pdf)) * light_color; // rt.h:258
                    } else {
                        const float pt_pdf = bsdf_pdf(sf, local_wo, local_wi) * G;
                        const float mis_weight = light_pdf / (light_pdf + pt_pdf);
                        sh_contrib = ((ps.thr * ((bsdf * G * cosine1) / light_pdf)) * mis_weight) * light_color; // rt.h:378
                    }
                    sh_d = sd; sh_tmax = light_distance - 0.001f; // origin = prd.position = ps.ro below
                    // an exactly-zero contribution (every hit on the glass lobe, whose evaluateBSDF is 0) cannot change L whatever
                    // the shadow ray returns (x + 0 == x): skip the trace.  A NaN contribution still goes through.
                    sh_valid = !(sh_contrib.x == 0.0f && sh_contrib.y == 0.0f && sh_contrib.z == 0.0f);
                }

                if (INTEGRATOR == HJR_INTEGRATOR_MIS_) { // BSDF-sampled light hit, rt.h:383-420
                    // MIS adds this term AFTER the NEE term of the same bounce (rt.h:378 then :414/:418); the NEE term is still
                    // pending, so resolve its shadow ray now to keep the order of the float additions
                    if (sh_valid) {
                        Hit shh;
                        Counters c; c.box = 0; c.tri = 0;
                        const bool occ = traverse<true, STATS, WIDTH, BLOCK, ST>(nodes, tris, prd.position, sh_d, 0.001f, sh_tmax, shh, stack, c);
                        if (STATS) { lc[2] += 1; lc[5] += c.box; lc[6] += c.tri; }
                        if (!occ) ps.L = ps.L + sh_contrib;
                        sh_valid = false;
                    }
                    float pt_pdf = 1.0f; // uninitialised in the reference when msGGX returns early; defined as 1
                    f3 local_wi = V(0.0f, 1.0f, 0.0f);
                    const f3 brdf = bsdf_sample(P, sf, local_wo, local_wi, pt_pdf, st);
                    const f3 wi = local_to_world(local_wi, t, n, b);
                    const float cosine1 = absdot(wi, n);
                    HitInfo lh;
                    ray_trace<STATS, AOVS, WIDTH, BLOCK, ST>(P, nodes, tris, mats, prd.position, wi, lh, stack, lc);
                    if (lh.is_hit) {
                        if (lh.is_light) {
                            const float cosine2 = absdot(-wi, lh.normal);
                            const float light_distance = length3(lh.position - prd.position);
                            const float invG = light_distance * light_distance / cosine2;
                            // getLightPDF(prim, inst) (light_sample.h:77-92): 1 / (area * light_prim_count), area from the light
                            // table's world vertices (== transform_position of the same object vertices); the table row of an
                            // emissive triangle is found by its global prim id (l4.w)
                            float lp = 0.0f;
                            if (!sf.is_specular) {
                                for (uint32_t li = 0; li < P.n_lights; li++) {
                                    const float4* Lr = lights + li * HJR_LIGHT_F4;
                                    if (f2bits(Lr[4].w) == lh.prim) {
                                        const float4 a0 = Lr[0], a1 = Lr[1], a2 = Lr[2];
                                        const f3 c = cross(V(a1.x, a1.y, a1.z) - V(a0.x, a0.y, a0.z), V(a2.x, a2.y, a2.z) - V(a0.x, a0.y, a0.z));
                                        const float area = length3(c) * 0.5f;
                                        lp = 1.0f / (area * P.n_lights);
                                        break;
                                    }
                                }
                                lp = lp * invG;
                            }
                            const float mis_weight = pt_pdf / (pt_pdf + lp);
                            ps.L = ps.L + ((((ps.thr * mis_weight) * cosine1) * lh.emission) * brdf) / pt_pdf; // rt.h:414
                        }
                    } else {
                        ps.L = ps.L + (((ps.thr * brdf) * cosine1) * lh.emission) / pt_pdf; // rt.h:418
                    }
                }

                HJR_TICK(3)
                float pdf = 1.0f;
                f3 local_wi = V(0.0f, 1.0f, 0.0f);
                if (INTEGRATOR != HJR_INTEGRATOR_PT_) (void)cmj_2d(st); // drawn and discarded by the reference (rt.h:266, 426)
                const f3 bsdf = bsdf_sample(P, sf, local_wo, local_wi, pdf, st);
                const f3 wi = local_to_world(local_wi, t, n, b);
                ps.thr = ps.thr * ((bsdf * fabsf(dot(wi, n))) / pdf); // rt.h:274
                ps.ro = prd.position;
                ps.rd = wi;
                fresh = false;
                ps.rng_depth = st.depth;
                ps.depth++;
                if (ps.depth == 10) { // MaxDepth (rt.h:166): the path is over; its last shadow ray, if any, is still pending
                    if (sh_valid) { fin_pending = true; close_sample(); }
                    else {
                        finish_sample(ps.L);
                        ps.L = V1(0.0f);
                        close_sample();
                        if (write_pending) write_out();
                    }
                }
                HJR_TICK(4)
            }
        }
        HJR_TICK(5)
    }
#ifdef HJR_TIMING
    if (lane == 0) { for (int i = 0; i < 6; i++) atomicAdd(&P.stats[HJR_NSTAT + i], tk[i]); atomicAdd(&P.stats[HJR_NSTAT + 6], tk6); atomicAdd(&P.stats[HJR_NSTAT + 7], tk7); }
    if (__ffsll((long long)__ballot(true)) - 1 == (int)lane) { for (int i = 0; i < 6; i++) atomicAdd(&P.stats[HJR_NSTAT + 8 + i], oc[i]); atomicAdd(&P.stats[HJR_NSTAT + 18], oc[6]); atomicAdd(&P.stats[HJR_NSTAT + 19], oc[7]); }
    if (lane == 0) for (int i = 0; i < 4; i++) atomicAdd(&P.stats[HJR_NSTAT + 14 + i], tx[i]);
#endif

    if (STATS) {
        for (int i = 0; i < HJR_NSTAT; i++) {
            unsigned long long v = lc[i];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
            if (lane == 0 && v) atomicAdd(&P.stats[i], v);
        }
    }
}

// ---- cost-ordered tile list.  The frame ends when the slowest work item ends, and an item (8 samples of one pixel, each up
// to 10 bounces, strictly sequential) can run for milliseconds: with plain scanline order the tail of the launch is whatever
// the last tiles happen to cost (5 ms on a 64 x 64 frame, 15 % of an 18 ms launch when the frame is split over 8 GPUs).
// One wave per owned tile casts the 64 pixel-centre rays (no RNG), classifies the tile by its costliest first hit
// (3 = glass, 2 = metallic, 1 = other surface, 0 = background or light) and the tiles are handed out class 3 first, background
// last: longest-processing-time-first scheduling, and waves whose lanes behave alike.  Only the ORDER of the work changes;
// every pixel is computed exactly as before.
template <int WIDTH>
__global__ void __launch_bounds__(64) hjr_classify_tiles_kernel(const KParams P)
{
    typedef LaneStack<uint32_t, 64, 0> ST;
    ST stack;
    stack.lds = reinterpret_cast<uint32_t*>(hjr_smem) + threadIdx.x;
    stack.spill = nullptr; stack.spill_stride = 0;
    uint32_t n_cls[4] = { 0u, 0u, 0u, 0u }; // per block; one atomic per class at the end (32 k atomics on four words cost 0.4 ms)
    for (uint32_t idx = blockIdx.x; idx < P.n_owned_tiles; idx += gridDim.x) {
        const uint32_t tile = idx * P.world + P.rank;
        const uint32_t tx = tile % P.tiles_x, ty = tile / P.tiles_x;
        const uint32_t px = tx * HJR_TILE + (threadIdx.x & 7u), py = ty * HJR_TILE + (threadIdx.x >> 3);
        uint32_t cls = 0;
        if (px < P.width && py < P.height) {
            const float W = (float)P.width, H = (float)P.height;
            const float u = (2.0f * ((float)px + 0.5f) - W) / H, v = (2.0f * ((float)py + 0.5f) - H) / H;
            const f3 cd = V(P.cam_dir[0], P.cam_dir[1], P.cam_dir[2]), cu = V(P.cam_up[0], P.cam_up[1], P.cam_up[2]);
            const f3 cr = V(P.cam_right[0], P.cam_right[1], P.cam_right[2]);
            const f3 d = normalize(cd * P.cam_f + cr * u + cu * v);
            Hit h;
            Counters cnt; cnt.box = cnt.tri = 0;
            if (traverse<false, false, WIDTH, 64, ST>(P.nodes, P.tri_geom, V(P.cam_pos[0], P.cam_pos[1], P.cam_pos[2]), d, 0.001f, 1e16f, h, stack, cnt)) {
                const float4* m = P.materials + f2bits(P.tri_geom[h.k * HJR_TRI_F4 + 2].z) * HJR_MAT_F4;
                const float4 m0 = m[0], m3 = m[3];
                cls = f2bits(m3.x) != 0 ? 0u : (f2bits(m3.y) != 0 ? 3u : (m0.w > 0.5f ? 2u : 1u));
            }
        }
        const uint32_t tcls = __ballot(cls == 3u) ? 3u : (__ballot(cls == 2u) ? 2u : (__ballot(cls == 1u) ? 1u : 0u));
        if (threadIdx.x == 0) P.tile_class[idx] = tcls;
        n_cls[0] += tcls == 0u; n_cls[1] += tcls == 1u; n_cls[2] += tcls == 2u; n_cls[3] += tcls == 3u;
    }
    if (threadIdx.x < 4u && n_cls[threadIdx.x]) atomicAdd(&P.tile_count[threadIdx.x], n_cls[threadIdx.x]);
}
__global__ void __launch_bounds__(256) hjr_order_tiles_kernel(const KParams P)
{
    const uint32_t idx = blockIdx.x * 256u + threadIdx.x;
    const bool live = idx < P.n_owned_tiles;
    const uint32_t cls = live ? P.tile_class[idx] : 0xffffffffu;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t pos = 0;
    for (uint32_t c = 0; c < 4u; c++) { // wave-aggregated scatter: one atomic per wave and class
        const unsigned long long m = __ballot(cls == c);
        if (m == 0ull) continue;
        uint32_t base = 0;
        if (lane == (uint32_t)(__ffsll((long long)m) - 1)) base = atomicAdd(&P.tile_count[4 + c], (uint32_t)__popcll(m));
        base = (uint32_t)__shfl((int)base, __ffsll((long long)m) - 1);
        if (cls == c) {
            uint32_t first = 0;
            for (uint32_t k = 3u; k > c; k--) first += P.tile_count[k];
            pos = first + base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        }
    }
    if (live) P.tile_order_w[pos] = idx * P.world + P.rank;
}

// From the second frame of a sequence on, the tiles are ordered by what they actually cost in the previous frame (closest-hit
// rays per sample) inside their first-hit class: a counting sort over 64 keys in two kernels; the order inside a key is arbitrary.
HD uint32_t cost_bucket(uint32_t cls, uint32_t cost, uint32_t cost_div)
{
    // key = (first-hit class, measured rays per sample in steps of 1/2): the class keeps waves of like materials together in
    // time (3 % at N = 1), the cost orders the tiles inside a class so that the last items of a class are its cheapest
    const uint32_t b = (uint32_t)(((unsigned long long)cost * 2ull) / cost_div);
    return (cls & 3u) * 16u + (b > 15u ? 15u : b);
}
__global__ void __launch_bounds__(256) hjr_cost_hist_kernel(const KParams P)
{
    __shared__ uint32_t h[64];
    if (threadIdx.x < 64u) h[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t idx = blockIdx.x * 256u + threadIdx.x;
    if (idx < P.n_owned_tiles) {
        const uint32_t b = cost_bucket(P.tile_class[idx], P.tile_cost[idx], P.cost_div);
        P.tile_bucket[idx] = b;
        atomicAdd(&h[b], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 64u && h[threadIdx.x]) atomicAdd(&P.cost_hist[threadIdx.x], h[threadIdx.x]);
}
__global__ void __launch_bounds__(256) hjr_cost_scatter_kernel(const KParams P)
{
    __shared__ uint32_t h[64], base[64];
    if (threadIdx.x < 64u) h[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t idx = blockIdx.x * 256u + threadIdx.x;
    const bool live = idx < P.n_owned_tiles;
    uint32_t b = 0, rank_in_block = 0;
    if (live) { b = P.tile_bucket[idx]; rank_in_block = atomicAdd(&h[b], 1u); P.tile_cost[idx] = 0u; } // zeroed for this frame's sums
    __syncthreads();
    if (threadIdx.x < 64u) {
        uint32_t first = 0;
        for (uint32_t k = 63u; k > threadIdx.x; k--) first += P.cost_hist[k]; // expensive buckets first
        base[threadIdx.x] = h[threadIdx.x] ? first + atomicAdd(&P.cost_hist[64u + threadIdx.x], h[threadIdx.x]) : 0u;
    }
    __syncthreads();
    if (live) P.tile_order_w[base[b] + rank_in_block] = idx * P.world + P.rank;
}

// Adds the chunk sums of every owned pixel in chunk order and scales by 1/spp: a fixed summation
// tree, so the frame is bitwise independent of which lane/wave/GPU rendered which chunk.  Streaming kernel: one lane
// per pixel, n_chunks coalesced float4 loads, one float4 store.
__global__ void __launch_bounds__(256) hjr_finalize_kernel(const KParams P)
{
    const size_t npix = (size_t)P.width * P.height;
    const float inv_spp = 1.0f / (float)P.spp;
    for (size_t pix = (size_t)blockIdx.x * blockDim.x + threadIdx.x; pix < npix; pix += (size_t)gridDim.x * blockDim.x) {
        const uint32_t x = (uint32_t)(pix % P.width), y = (uint32_t)(pix / P.width);
        const uint32_t tile = (y / HJR_TILE) * P.tiles_x + (x / HJR_TILE);
        if (tile % P.world != P.rank) continue;
        float4 a = make_float4(0.0f, 0.0f, 0.0f, 0.0f), b = a, c = a;
        for (uint32_t k = 0; k < P.n_chunks; k++) {
            const float4 v = P.part_color[(size_t)k * npix + pix];
            a.x = a.x + v.x; a.y = a.y + v.y; a.z = a.z + v.z;
            if (P.aov_albedo) { const float4 w = P.part_albedo[(size_t)k * npix + pix]; b.x = b.x + w.x; b.y = b.y + w.y; b.z = b.z + w.z; }
            if (P.aov_normal) { const float4 w = P.part_normal[(size_t)k * npix + pix]; c.x = c.x + w.x; c.y = c.y + w.y; c.z = c.z + w.z; }
        }
        P.aov_color[pix] = make_float4(a.x * inv_spp, a.y * inv_spp, a.z * inv_spp, 1.0f);
        if (P.aov_albedo) P.aov_albedo[pix] = make_float4(b.x * inv_spp, b.y * inv_spp, b.z * inv_spp, 1.0f);
        if (P.aov_normal) P.aov_normal[pix] = make_float4(c.x * inv_spp, c.y * inv_spp, c.z * inv_spp, 1.0f);
    }
}
