/*
 * hjr_oracle.c — CPU ORACLE (test infrastructure; see hjr_oracle.h for the rules of use).
 *
 * Every function names the reference lines it restates (paths relative to /root/reference/include).
 * Arithmetic rules: fp32 throughout, evaluated left to right exactly as the
 * reference writes it, NO fused contraction (build with -ffp-contract=off) except where the
 * reference itself calls fma() or where the code is build-defined (ray/triangle test) and says fmaf().
 * Vector helpers restate NVIDIA OptiX SDK 7.7 sutil/vec_math.h (third-party, un-vendored in the
 * reference: float3/float == multiply by 1.0f/s, normalize == v * (1.0f/sqrtf(dot)) ...).
 */
#include "hjr_oracle.h"
#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

typedef struct { float x, y, z; } f3;
typedef struct { float x, y; } f2;

/* ------------------------------------------------------------------ sutil/vec_math.h restated */
static inline f3 V(float x, float y, float z) { f3 r = { x, y, z }; return r; }
static inline f3 V1(float s) { return V(s, s, s); }
static inline f3 add(f3 a, f3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline f3 sub(f3 a, f3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline f3 mul(f3 a, f3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline f3 muls(f3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
static inline f3 divs(f3 a, float s) { float inv = 1.0f / s; return muls(a, inv); }
static inline f3 neg(f3 a) { return V(-a.x, -a.y, -a.z); }
static inline f3 ssub(float s, f3 a) { return V(s - a.x, s - a.y, s - a.z); }
static inline float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline f3 cross(f3 a, f3 b) {
    return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline float length3(f3 v) { return sqrtf(dot(v, v)); }
static inline f3 normalize(f3 v) { float invLen = 1.0f / sqrtf(dot(v, v)); return muls(v, invLen); }
static inline f3 reflect3(f3 i, f3 n) { return sub(i, muls(muls(n, 2.0f), dot(n, i))); }
static inline f3 lerp3(f3 a, f3 b, float t) { return add(a, muls(sub(b, a), t)); }
static inline float clampf(float f, float a, float b) { return fmaxf(a, fminf(f, b)); }

#define HJ_PI 3.14159265358979323846f
#define HJ_PI2 6.28318530717958647692f
#define HJ_INV_PI 0.31830988618379067154f

/* ------------------------------------------------------------------ portable transcendental set
 * Built from IEEE + - * / sqrt fma floor and integer bit operations only, so the HIP kernel's copy
 * (henjou-renderer_amd/csrc/hjr_math.hip.h) produces the same bits.  Polynomials: Cephes single precision. */
static inline float bits2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t f2bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

static inline void p_sincos(float x, float* s, float* c)
{
    float fj = floorf(x * 0.636619772367581343f + 0.5f);
    int j = (int)fj;
    float r = fmaf(fj, -1.5703125f, x);
    r = fmaf(fj, -4.837512969970703125e-4f, r);
    r = fmaf(fj, -7.54978995489188216e-8f, r);
    float z = r * r;
    float sp = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f) * z, r, r);
    float cp = fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f),
                    z * z, fmaf(-0.5f, z, 1.0f));
    switch (j & 3) {
    case 0: *s = sp; *c = cp; break;
    case 1: *s = cp; *c = -sp; break;
    case 2: *s = -sp; *c = -cp; break;
    default: *s = -cp; *c = sp; break;
    }
}
float hjo_p_sin(float x) { float s, c; p_sincos(x, &s, &c); return s; }
float hjo_p_cos(float x) { float s, c; p_sincos(x, &s, &c); return c; }

static inline float p_asin_poly(float x, float z)
{
    float p = fmaf(fmaf(fmaf(fmaf(4.2163199048e-2f, z, 2.4181311049e-2f), z, 4.5470025998e-2f), z,
                        7.4953002686e-2f), z, 1.6666752422e-1f);
    return fmaf(p * z, x, x);
}
float hjo_p_acos(float x)
{
    if (x > 0.5f) {
        float z = 0.5f * (1.0f - x);
        float s = sqrtf(z);
        return 2.0f * p_asin_poly(s, z);
    }
    if (x < -0.5f) {
        float z = 0.5f * (1.0f + x);
        float s = sqrtf(z);
        return HJ_PI - 2.0f * p_asin_poly(s, z);
    }
    return 1.57079632679489661923f - p_asin_poly(x, x * x);
}

static inline float p_log(float x) /* x > 0, finite, normal */
{
    uint32_t u = f2bits(x);
    int e = (int)(u >> 23) - 126;
    float m = bits2f((u & 0x007fffffu) | 0x3f000000u); /* [0.5,1) */
    if (m < 0.707106781186547524f) { e -= 1; m = m + m - 1.0f; } else { m = m - 1.0f; }
    float z = m * m;
    float y = fmaf(7.0376836292e-2f, m, -1.1514610310e-1f);
    y = fmaf(y, m, 1.1676998740e-1f);
    y = fmaf(y, m, -1.2420140846e-1f);
    y = fmaf(y, m, 1.4249322787e-1f);
    y = fmaf(y, m, -1.6668057665e-1f);
    y = fmaf(y, m, 2.0000714765e-1f);
    y = fmaf(y, m, -2.4999993993e-1f);
    y = fmaf(y, m, 3.3333331174e-1f);
    y = y * m * z;
    float fe = (float)e;
    y = fmaf(fe, -2.12194440e-4f, y);
    y = fmaf(-0.5f, z, y);
    float r = m + y;
    return fmaf(fe, 0.693359375f, r);
}
static inline float p_exp(float x) /* -87 <= x <= 88.7 */
{
    float fn = floorf(fmaf(1.44269504088896341f, x, 0.5f));
    int n = (int)fn;
    float r = fmaf(fn, -0.693359375f, x);
    r = fmaf(fn, 2.12194440e-4f, r);
    float z = r * r;
    float p = fmaf(1.9875691500e-4f, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    p = fmaf(p, z, r) + 1.0f;
    if (n > 127) { p = p * 1.70141183460469231732e38f; n -= 127; }
    if (n < -126) return 0.0f;
    return p * bits2f((uint32_t)(n + 127) << 23);
}
float hjo_p_pow(float x, float y)
{
    if (y == 0.0f) return 1.0f;
    if (x == 1.0f) return 1.0f;
    if (x != x || y != y) return x + y;
    if (x < 0.0f) return bits2f(0x7fc00000u);
    if (x == 0.0f) return (y > 0.0f) ? 0.0f : bits2f(0x7f800000u);
    if (x > FLT_MAX) return (y > 0.0f) ? x : 0.0f;
    float lx = (x < FLT_MIN) ? p_log(x * 16777216.0f) - 16.6355323334f : p_log(x); /* denormal base: rescale by 2^24 */
    float t = y * lx;
    if (t != t) return t;
    if (t > 88.7f) return bits2f(0x7f800000u);
    if (t < -87.0f) return 0.0f;
    return p_exp(t);
}
float hjo_p_pow5(float x) { float x2 = x * x; float x4 = x2 * x2; return x4 * x; }
/* Cephes atanf on [0, inf) + quadrant logic; x == y == 0 -> 0 */
static inline float p_atan_pos(float x)
{
    float y0;
    if (x > 2.414213562373095f) { y0 = 1.57079632679489661923f; x = -(1.0f / x); }
    else if (x > 0.4142135623730950f) { y0 = 0.78539816339744830962f; x = (x - 1.0f) / (x + 1.0f); }
    else y0 = 0.0f;
    float z = x * x;
    float p = fmaf(fmaf(fmaf(8.05374449538e-2f, z, -1.38776856032e-1f), z, 1.99777106478e-1f), z, -3.33329491539e-1f);
    return y0 + fmaf(p * z, x, x);
}
float hjo_p_atan2(float y, float x)
{
    if (x != x || y != y) return x + y;
    if (y == 0.0f) return (x >= 0.0f && !(f2bits(x) >> 31)) ? y : copysignf(HJ_PI, y);
    if (x == 0.0f) return copysignf(1.57079632679489661923f, y);
    float a = p_atan_pos(fabsf(y / x));
    if (x < 0.0f) a = HJ_PI - a;
    return copysignf(a, y);
}

/* math back-end dispatch.
 * mode 0 (LIBM)      : the reference's un-suffixed sin/cos/acos/pow/sqrt/fma calls bind to the FLOAT overloads,
 *                      as they do under nvcc (CUDA math API) — glibc sinf/cosf/acosf/powf.
 * mode 1 (PORTABLE)  : same binding, own transcendental functions (bit-identical on CPU and gfx950).
 * mode 2 (HOSTF64)   : what a g++ host compile of the reference headers computes: with only <cmath> in scope the
 *                      un-suffixed names bind to ::sin(double) etc., so those calls (and the products they sit in)
 *                      are evaluated in double.  SURVEY.md §8c's known-answer values were produced that way;
 *                      this mode exists only to reproduce them bit for bit (tests/test_oracle_kat.py). */
static inline float m_sin(int mode, float x) { return mode == 1 ? hjo_p_sin(x) : (mode == 2 ? (float)sin((double)x) : sinf(x)); }
static inline float m_cos(int mode, float x) { return mode == 1 ? hjo_p_cos(x) : (mode == 2 ? (float)cos((double)x) : cosf(x)); }
static inline float m_powf(int mode, float x, float y) { return mode == 1 ? hjo_p_pow(x, y) : powf(x, y); } /* powf(): suffixed in the reference */
static inline float m_pow5(int mode, float x) { return mode == 1 ? hjo_p_pow5(x) : (mode == 2 ? (float)pow((double)x, 5.0) : powf(x, 5.0f)); } /* pow(x,5.0f) */
/* `cos(phi) * s` / `sin(phi) * s` */
static inline float m_cos_mul(int mode, float phi, float s) { return mode == 2 ? (float)(cos((double)phi) * (double)s) : m_cos(mode, phi) * s; }
static inline float m_sin_mul(int mode, float phi, float s) { return mode == 2 ? (float)(sin((double)phi) * (double)s) : m_sin(mode, phi) * s; }
/* `0.5f * acos(x)` */
static inline float m_half_acos(int mode, float x)
{
    return mode == 1 ? 0.5f * hjo_p_acos(x) : (mode == 2 ? (float)(0.5 * acos((double)x)) : 0.5f * acosf(x));
}
/* `fma(a, b, c)` */
static inline float m_fma(int mode, float a, float b, float c) { return mode == 2 ? (float)fma((double)a, (double)b, (double)c) : fmaf(a, b, c); }
/* logf(alpha2) in clearcoat_D is only ever taken of the constant 1e-6f (disneyBRDF.h:138,175). */
#define HJ_LOG_CLEARCOAT_ALPHA2 (-13.8155105579642741f)

/* ------------------------------------------------------------------ cmj.h */
typedef struct { uint64_t n_spp; uint32_t scramble, depth, image_idx; } cmj_state; /* cmj.h:53-58 */

uint32_t hjo_xxhash32_u4(uint32_t px, uint32_t py, uint32_t pz, uint32_t pw) /* cmj.h:38-51 */
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
uint32_t hjo_cmj_permute(uint32_t i, uint32_t l, uint32_t p) /* cmj.h:60-91 */
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
float hjo_cmj_randfloat(uint32_t i, uint32_t p) /* cmj.h:93-106 */
{
    i ^= p;
    i ^= i >> 17; i ^= i >> 10; i *= 0xb36534e5;
    i ^= i >> 12; i ^= i >> 21; i *= 0x93fc4795;
    i ^= 0xdf6e307f;
    i ^= i >> 17; i *= 1 | p >> 18;
    return i * (1.0f / 4294967808.0f);
}
static inline f2 cmj(uint32_t index, uint32_t scramble) /* cmj.h:108-117, CMJ_M = CMJ_N = 4 */
{
    index = hjo_cmj_permute(index, 16, scramble * 0x51633e2d);
    uint32_t sx = hjo_cmj_permute(index % 4, 4, scramble * 0xa511e9b3);
    uint32_t sy = hjo_cmj_permute(index / 4, 4, scramble * 0x63d83595);
    float jx = hjo_cmj_randfloat(index, scramble * 0xa399d265);
    float jy = hjo_cmj_randfloat(index, scramble * 0x711ad6a5);
    f2 r;
    r.x = (index % 4 + (sy + jx) / 4) / 4;
    r.y = (index / 4 + (sx + jy) / 4) / 4;
    return r;
}
void hjo_cmj(uint32_t index, uint32_t scramble, float* o) { f2 r = cmj(index, scramble); o[0] = r.x; o[1] = r.y; }
static inline f2 cmj_2d(cmj_state* st) /* cmj.h:119-128 */
{
    const uint32_t index = (uint32_t)(st->n_spp % 16);
    const uint32_t scramble = hjo_xxhash32_u4((uint32_t)(st->n_spp / 16), st->image_idx, st->depth, st->scramble);
    f2 r = cmj(index, scramble);
    st->depth++;
    return r;
}
static inline float cmj_1d(cmj_state* st) { return cmj_2d(st).x; } /* cmj.h:130-133 */
void hjo_cmj_2d(uint32_t* s5, float* o)
{
    cmj_state st = { (uint64_t)s5[0] | ((uint64_t)s5[1] << 32), s5[2], s5[3], s5[4] };
    f2 r = cmj_2d(&st);
    s5[3] = st.depth;
    o[0] = r.x; o[1] = r.y;
}
static inline cmj_state state_from5(const uint32_t* s5)
{
    cmj_state st = { (uint64_t)s5[0] | ((uint64_t)s5[1] << 32), s5[2], s5[3], s5[4] };
    return st;
}

/* ------------------------------------------------------------------ kernel/math.h */
static inline f3 cosine_sampling(int mm, float u, float v, float* pdf) /* math.h:7-15 */
{
    float phi = 2.0f * HJ_PI * v;
    float theta = m_half_acos(mm, 1.0f - 2.0f * u);
    float cosTheta = m_cos(mm, theta);
    float sinTheta = m_sin(mm, theta);
    *pdf = cosTheta / HJ_PI;
    return V(m_cos_mul(mm, phi, sinTheta), cosTheta, m_sin_mul(mm, phi, sinTheta));
}
void hjo_cosine_sampling(int mm, float u, float v, float* wi, float* pdf)
{
    f3 w = cosine_sampling(mm, u, v, pdf); wi[0] = w.x; wi[1] = w.y; wi[2] = w.z;
}
static inline f3 schlick3(int mm, f3 F0, f3 w, f3 n) /* math.h:26-29 */
{
    float term1 = 1.0f - dot(w, n);
    return add(muls(ssub(1.0f, F0), m_pow5(mm, term1)), F0);
}
static inline float schlick_ior(int mm, float no, float ni, f3 w, f3 n) /* math.h:31-37 */
{
    float F0 = (no - ni) / (no + ni);
    F0 = F0 * F0;
    float term1 = 1.0f - dot(w, n);
    return F0 + (1.0f - F0) * m_pow5(mm, term1);
}
float hjo_schlick_ior(int mm, float no, float ni, const float* w, const float* n)
{
    return schlick_ior(mm, no, ni, V(w[0], w[1], w[2]), V(n[0], n[1], n[2]));
}
static inline void orthonormal_basis(f3 n, f3* t, f3* b) /* math.h:43-51 */
{
    float sign = copysignf(1.0f, n.z);
    const float a = -1.0f / (sign + n.z);
    const float bb = n.x * n.y * a;
    *t = V(1.0f + sign * n.x * n.x * a, sign * bb, -sign * n.x);
    *b = V(bb, sign + n.y * n.y * a, -n.y);
}
void hjo_orthonormal_basis(const float* n, float* t, float* b)
{
    f3 tt, bb; orthonormal_basis(V(n[0], n[1], n[2]), &tt, &bb);
    t[0] = tt.x; t[1] = tt.y; t[2] = tt.z; b[0] = bb.x; b[1] = bb.y; b[2] = bb.z;
}
static inline f3 world_to_local(f3 v, f3 t, f3 n, f3 b) { return V(dot(v, t), dot(v, n), dot(v, b)); } /* math.h:53-59 */
static inline f3 local_to_world(f3 v, f3 t, f3 n, f3 b) /* math.h:61-71 */
{
    return V(v.x * t.x + v.y * n.x + v.z * b.x, v.x * t.y + v.y * n.y + v.z * b.y, v.x * t.z + v.y * n.z + v.z * b.z);
}
/* Matrix4x3 rows r0,r1,r2 as 12 floats (cu/matrix_4x3.h:12-16); dot(float4,float4) of sutil. */
static inline f3 transform_position(const float* m, f3 p) /* math.h:73-76 */
{
    return V(m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3] * 1.0f,
             m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7] * 1.0f,
             m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11] * 1.0f);
}
static inline f3 transform_normal(const float* m, f3 n) /* math.h:78-87 (transposed rows, w = 0) */
{
    return V(m[0] * n.x + m[4] * n.y + m[8] * n.z + 0.0f * 0.0f,
             m[1] * n.x + m[5] * n.y + m[9] * n.z + 0.0f * 0.0f,
             m[2] * n.x + m[6] * n.y + m[10] * n.z + 0.0f * 0.0f);
}
static inline float norm2(f3 v) { return v.x * v.x + v.y * v.y + v.z * v.z; } /* math.h:88-90 */
static inline int refract3(f3 v, f3 n, float ior1, float ior2, f3* r) /* math.h:92-103 */
{
    const f3 t_h = muls(sub(v, muls(n, dot(v, n))), -ior1 / ior2);
    if (norm2(t_h) > 1.0f) return 0;
    const f3 t_p = muls(n, -sqrtf(fmaxf(1.0f - norm2(t_h), 0.0f)));
    *r = add(t_h, t_p);
    return 1;
}
int hjo_refract(const float* v, const float* n, float i1, float i2, float* r)
{
    f3 rr = V(0, 0, 0); int ok = refract3(V(v[0], v[1], v[2]), V(n[0], n[1], n[2]), i1, i2, &rr);
    r[0] = rr.x; r[1] = rr.y; r[2] = rr.z; return ok;
}
static inline float absdot(f3 a, f3 b) { return fabsf(dot(a, b)); } /* math.h:105-107 */

/* ------------------------------------------------------------------ Payload.h:12-42 */
typedef struct {
    int is_hit;
    f3 position, normal;
    f2 texcoord;
    f3 basecolor;
    float metallic, roughness, sheen, clearcoat, ior, transmission;
    int is_specular;
    f3 emission;
    int is_light, is_thinfilm;
    int primitive_id, instance_id;
} payload;

static inline void payload_init(payload* p)
{
    memset(p, 0, sizeof(*p));
    p->ior = 1.0f; p->transmission = 1.0f;
}

/* ------------------------------------------------------------------ thin-film LUT (disneyBRDF.h:11-14)
 * Sampler state from renderer.h:854-898: uchar4 read as normalised float, linear filter, wrap,
 * normalised coordinates.  Filtering follows the CUDA programming-guide formula (texel-centre offset,
 * weights in 1.8 fixed point); the exact hardware rounding is unobservable here => build-defined. */
void hjo_lut_fetch(const uint8_t* rgba, int w, int h, float u, float v, float* out)
{
    if (!rgba || w <= 0 || h <= 0) { out[0] = out[1] = out[2] = 0.0f; return; }
    float x = u * (float)w - 0.5f, y = v * (float)h - 0.5f;
    float fx = floorf(x), fy = floorf(y);
    float ax = floorf((x - fx) * 256.0f + 0.5f) * (1.0f / 256.0f);
    float ay = floorf((y - fy) * 256.0f + 0.5f) * (1.0f / 256.0f);
    int i0 = (int)fx % w; if (i0 < 0) i0 += w;
    int j0 = (int)fy % h; if (j0 < 0) j0 += h;
    int i1 = (i0 + 1) % w, j1 = (j0 + 1) % h;
    for (int c = 0; c < 3; c++) {
        float t00 = (float)rgba[4 * (j0 * w + i0) + c] * (1.0f / 255.0f);
        float t10 = (float)rgba[4 * (j0 * w + i1) + c] * (1.0f / 255.0f);
        float t01 = (float)rgba[4 * (j1 * w + i0) + c] * (1.0f / 255.0f);
        float t11 = (float)rgba[4 * (j1 * w + i1) + c] * (1.0f / 255.0f);
        out[c] = (1.0f - ax) * (1.0f - ay) * t00 + ax * (1.0f - ay) * t10 + (1.0f - ax) * ay * t01 + ax * ay * t11;
    }
}

/* ------------------------------------------------------------------ material textures (renderer.h:740-800) and the equirect sky
 * (renderer.h:802-851).  Sampling in the closest-hit / miss programs is build-defined (their source is missing):
 * 8-bit RGBA, wrap, bilinear with the CUDA 1.8 fixed-point weights, sRGB -> linear per texel BEFORE filtering when the texture
 * is TexType::sRGB; sky: float texels, same filter, direction -> (u, v) = (atan2(d.z, d.x) / 2pi + 0.5, acos(d.y) / pi). */
static float g_srgb_lut[256];
static int g_srgb_ready = 0;
static void srgb_init(void)
{
    if (g_srgb_ready) return;
    for (int i = 0; i < 256; i++) {
        double c = (double)i / 255.0;
        g_srgb_lut[i] = (float)(c <= 0.04045 ? c / 12.92 : pow((c + 0.055) / 1.055, 2.4));
    }
    g_srgb_ready = 1;
}
void hjo_tex_fetch(const uint8_t* rgba, int w, int h, int srgb, float u, float v, float* out)
{
    if (!rgba || w <= 0 || h <= 0) { out[0] = out[1] = out[2] = 0.0f; return; }
    srgb_init();
    float x = u * (float)w - 0.5f, y = v * (float)h - 0.5f;
    float fx = floorf(x), fy = floorf(y);
    float ax = floorf((x - fx) * 256.0f + 0.5f) * (1.0f / 256.0f);
    float ay = floorf((y - fy) * 256.0f + 0.5f) * (1.0f / 256.0f);
    int i0 = (int)fx % w; if (i0 < 0) i0 += w;
    int j0 = (int)fy % h; if (j0 < 0) j0 += h;
    int i1 = (i0 + 1) % w, j1 = (j0 + 1) % h;
    float w00 = (1.0f - ax) * (1.0f - ay), w10 = ax * (1.0f - ay), w01 = (1.0f - ax) * ay, w11 = ax * ay;
    for (int c = 0; c < 3; c++) {
        uint8_t b00 = rgba[4 * (j0 * w + i0) + c], b10 = rgba[4 * (j0 * w + i1) + c], b01 = rgba[4 * (j1 * w + i0) + c], b11 = rgba[4 * (j1 * w + i1) + c];
        float t00, t10, t01, t11;
        if (srgb) { t00 = g_srgb_lut[b00]; t10 = g_srgb_lut[b10]; t01 = g_srgb_lut[b01]; t11 = g_srgb_lut[b11]; }
        else { const float k = 1.0f / 255.0f; t00 = (float)b00 * k; t10 = (float)b10 * k; t01 = (float)b01 * k; t11 = (float)b11 * k; }
        out[c] = w00 * t00 + w10 * t10 + w01 * t01 + w11 * t11;
    }
}
static inline float m_atan2(int mode, float y, float x) { return mode == 1 ? hjo_p_atan2(y, x) : atan2f(y, x); }
static inline float m_acos1(int mode, float x) { return mode == 1 ? hjo_p_acos(x) : acosf(x); }
void hjo_sky_fetch(int mm, const float* rgba, int w, int h, const float* d, float* out)
{
    float u = m_atan2(mm, d[2], d[0]) * 0.15915494309189533577f + 0.5f;
    float v = m_acos1(mm, clampf(d[1], -1.0f, 1.0f)) * HJ_INV_PI;
    float x = u * (float)w - 0.5f, y = v * (float)h - 0.5f;
    float fx = floorf(x), fy = floorf(y);
    float ax = floorf((x - fx) * 256.0f + 0.5f) * (1.0f / 256.0f);
    float ay = floorf((y - fy) * 256.0f + 0.5f) * (1.0f / 256.0f);
    int i0 = (int)fx % w; if (i0 < 0) i0 += w;
    int j0 = (int)fy % h; if (j0 < 0) j0 += h;
    int i1 = (i0 + 1) % w, j1 = (j0 + 1) % h;
    float w00 = (1.0f - ax) * (1.0f - ay), w10 = ax * (1.0f - ay), w01 = (1.0f - ax) * ay, w11 = ax * ay;
    for (int c = 0; c < 3; c++)
        out[c] = w00 * rgba[4 * (j0 * w + i0) + c] + w10 * rgba[4 * (j0 * w + i1) + c] + w01 * rgba[4 * (j1 * w + i0) + c] + w11 * rgba[4 * (j1 * w + i1) + c];
}

/* ------------------------------------------------------------------ DisneyBRDF (disneyBRDF.h:16-327) */
typedef struct {
    f3 basecolor;
    float alpha, metallic, sheen, clearcoat, clearcoatAlpha, subsurface;
    int is_thinfilm;
    const uint8_t* lut; int lut_w, lut_h;
    int mm;
} disney;

static inline void disney_init(disney* d, const payload* p, int mm, const uint8_t* lut, int lw, int lh) /* :165-177 */
{
    d->basecolor = p->basecolor;
    d->alpha = clampf(p->roughness * p->roughness, 0.01f, 1.0f);
    d->subsurface = 0.0f;
    d->metallic = p->metallic;
    d->sheen = p->sheen;
    d->clearcoat = p->clearcoat;
    { const float a = 0.1f, b = 0.001f, t = 1.0f; d->clearcoatAlpha = (1 - t) * a + t * b; } /* math.h:109-111 */
    d->is_thinfilm = p->is_thinfilm;
    d->lut = lut; d->lut_w = lw; d->lut_h = lh; d->mm = mm;
}
static inline float d_GGX_D(const disney* d, f3 wm) /* :44-48 */
{
    float a = d->alpha;
    float term1 = wm.x * wm.x / (a * a) + wm.z * wm.z / (a * a) + wm.y * wm.y;
    float term2 = HJ_PI * a * a * term1 * term1;
    return 1.0f / term2;
}
static inline float d_Lambda(const disney* d, f3 w) /* :58-61 */
{
    float a = d->alpha;
    float delta = 1.0f + (a * a * w.x * w.x + a * a * w.z * w.z) / (w.y * w.y);
    return (-1.0f + sqrtf(delta)) * 0.5f;
}
static inline float d_G1(const disney* d, f3 w) { return 1.0f / (1.0f + d_Lambda(d, w)); } /* :50-52 */
static inline float d_G2(const disney* d, f3 wi, f3 wo) { return 1.0f / (1.0f + d_Lambda(d, wi) + d_Lambda(d, wo)); } /* :54-56 */
static inline float d_getPDFDiffuse(f3 wi) { return fabsf(wi.y) * HJ_INV_PI; } /* :40-42 */
static inline f3 d_sampleDiffuse(const disney* d, f2 uv, float* pdf) /* :30-38 */
{
    int mm = d->mm;
    float theta = m_half_acos(mm, 1.0f - 2.0f * uv.x);
    float phi = 2.0f * HJ_PI * uv.y;
    float cosTheta = m_cos(mm, theta);
    float sinTheta = m_sin(mm, theta);
    f3 wi = V(m_cos_mul(mm, phi, sinTheta), cosTheta, m_sin_mul(mm, phi, sinTheta));
    *pdf = d_getPDFDiffuse(wi);
    return wi;
}
/* spherical-cap VNDF sampling, shared by Disney (:64-80) and msGGX (BSDFs.h:616-632) */
static inline f3 sample_visible_normal(int mm, float alpha, f2 uv, f3 wo)
{
    f3 strech_wo = normalize(V(wo.x * alpha, wo.y, wo.z * alpha));
    float phi = 2.0f * HJ_PI * uv.x;
    float z = m_fma(mm, (1.0f - uv.y), (1.0f + strech_wo.y), -strech_wo.y);
    float sinTheta = sqrtf(clampf(1.0f - z * z, 0.0f, 1.0f));
    float x = m_cos_mul(mm, phi, sinTheta);
    float y = m_sin_mul(mm, phi, sinTheta);
    f3 c = V(x, z, y);
    f3 h = add(c, strech_wo);
    return normalize(V(h.x * alpha, h.y, h.z * alpha));
}
static inline float d_getPDFSpecular(const disney* d, f3 wm, f3 wo) /* :88-90 */
{
    return 0.25f * d_GGX_D(d, wm) * d_G1(d, wo) * absdot(wo, wm) / (absdot(wm, wo) * fabsf(wo.y));
}
static inline float clearcoat_D(f3 wm, float alpha) /* :131-139 */
{
    float alpha2 = alpha * alpha;
    float t = 1.0f + (alpha2 - 1.0f) * wm.y * wm.y;
    return (alpha2 - 1.0f) / (HJ_PI * HJ_LOG_CLEARCOAT_ALPHA2 * t);
}
static inline float d_getPDFClearcoat(const disney* d, f3 wm, f3 wo) /* :102-104 */
{
    return clearcoat_D(wm, d->clearcoatAlpha) * fabsf(wm.y) / (4.0f * fabsf(dot(wm, wo)));
}
static inline f3 d_sampleClearcoat(const disney* d, f2 uv, f3 wo, float* pdf) /* :93-100 */
{
    int mm = d->mm;
    float ca = d->clearcoatAlpha;
    float cosineTheta = sqrtf(fmaxf((1.0f - m_powf(mm, ca * ca, 1.0f - uv.x)) / (1.0f - ca * ca), 0.0f));
    float sinTheta = sqrtf(fmaxf(1.0f - cosineTheta * cosineTheta, 0.0f));
    float phi = HJ_PI2 * uv.y;
    f3 wm = V(m_cos_mul(mm, phi, sinTheta), cosineTheta, m_sin_mul(mm, phi, sinTheta));
    *pdf = d_getPDFClearcoat(d, wm, wo);
    return wm;
}
static inline float f_tSchlick(float wn, float F90) /* :106-109 */
{
    float delta = fmaxf(1.0f - wn, 0.0f);
    return 1.0f + (F90 - 1.0f) * delta * delta * delta * delta * delta;
}
static inline f3 d_specular(const disney* d, f3 wo, f3 wi, f3 F0) /* :112-120 */
{
    f3 wm = normalize(add(wo, wi));
    float ggxD = d_GGX_D(d, wm);
    float ggxG = d_G2(d, wi, wo);
    f3 ggxF = schlick3(d->mm, F0, wo, wm);
    return divs(muls(muls(muls(ggxF, 0.25f), ggxD), ggxG), fabsf(wo.y) * fabsf(wi.y));
}
static inline float clearcoat_Lambda(f3 w, float alpha) /* :126-129 */
{
    float term1 = 1.0f + (alpha * alpha * w.x * w.x + alpha * alpha * w.z * w.z) / (w.y * w.y);
    return 0.5f * (-1.0f + sqrtf(term1));
}
static inline f3 d_clearcoat(const disney* d, f3 wo, f3 wi, float clearcoat_alpha) /* :142-150 */
{
    const f3 wm = normalize(add(wo, wi));
    float cD = clearcoat_D(wm, clearcoat_alpha);
    float cG = 1.0f / (1.0f + clearcoat_Lambda(wi, 0.25f) + clearcoat_Lambda(wo, 0.25f));
    f3 cF = schlick3(d->mm, V1(0.04f), wo, wm);
    return divs(muls(cF, 0.25f * cD * cG), fabsf(wo.y) * fabsf(wi.y));
}
static inline f3 disney_eval(const disney* d, f3 wo, f3 wi) /* :179-235 */
{
    f3 wm = normalize(add(wo, wi));
    float dot_wi_n = fabsf(wi.y);
    float dot_wo_n = fabsf(wi.y); /* sic: the reference uses wi here (:189) */
    float cosine_d = absdot(wi, wm);
    float F_D90 = 0.5f + 2.0f * d->alpha * cosine_d * cosine_d;
    float f_tsi = f_tSchlick(dot_wi_n, F_D90);
    float f_tso = f_tSchlick(dot_wo_n, F_D90);
    f3 f_diffuse = muls(muls(muls(d->basecolor, f_tsi), f_tso), HJ_INV_PI);
    float deltacos = 1.0f / (dot_wi_n + dot_wo_n) - 0.5f;
    f3 f_subsurface = muls(muls(muls(d->basecolor, HJ_INV_PI), 1.25f), (f_tsi * f_tso * deltacos + 0.5f));
    f3 F0 = lerp3(V1(0.08f), d->basecolor, d->metallic);
    if (d->is_thinfilm) {
        float thickness = d->basecolor.x;
        float cosine = absdot(wi, wm);
        float l[3]; hjo_lut_fetch(d->lut, d->lut_w, d->lut_h, thickness, cosine, l);
        F0 = V(l[0], l[1], l[2]);
    }
    f3 f_specular = d_specular(d, wo, wi, F0);
    float delta = fmaxf(1.0f - absdot(wi, wm), 0.0f);
    f3 f_sheen = muls(muls(muls(muls(muls(muls(V1(1.0f), d->sheen), delta), delta), delta), delta), delta);
    f3 f_clearcoat = muls(d_clearcoat(d, wo, wi, d->clearcoatAlpha), 0.25f);
    return add(add(muls(add(lerp3(f_diffuse, f_subsurface, d->subsurface), f_sheen), (1.0f - d->metallic)), f_specular),
               muls(f_clearcoat, d->clearcoat));
}
static inline f3 disney_sample(const disney* d, f3 wo, f3* wi, float* pdf, cmj_state* st) /* :237-307 */
{
    float diffuseWeight = 1.0f * (1.0f - d->metallic);
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
        *wi = d_sampleDiffuse(d, xi, &pdf_diffuse);
        f3 wm = normalize(add(*wi, wo));
        pdf_specular = d_getPDFSpecular(d, wm, wo);
        pdf_clearcoat = d_getPDFClearcoat(d, wm, wo);
    } else if (select_p < dw + sw) {
        f3 wm = sample_visible_normal(d->mm, d->alpha, xi, wo);
        pdf_specular = d_getPDFSpecular(d, wm, wo);
        *wi = reflect3(neg(wo), wm);
        pdf_diffuse = d_getPDFDiffuse(*wi);
        pdf_clearcoat = d_getPDFClearcoat(d, wm, wo);
    } else {
        f3 wm = d_sampleClearcoat(d, xi, wo, &pdf_clearcoat);
        *wi = reflect3(neg(wo), wm);
        pdf_diffuse = d_getPDFDiffuse(*wi);
        pdf_specular = d_getPDFSpecular(d, wm, wo);
    }
    *pdf = dw * pdf_diffuse + sw * pdf_specular + cw * pdf_clearcoat;
    if (wi->y < 0.0f) { *pdf = 1.0f; return V1(0.0f); }
    return disney_eval(d, wo, *wi);
}
static inline float disney_pdf(const disney* d, f3 wo, f3 wi) /* :309-326 */
{
    float diffuseWeight = 1.0f * (1.0f - d->metallic);
    float specularWeight = 0.5f, clearcoatWeight = 0.0f;
    float sumWeight = diffuseWeight + specularWeight + clearcoatWeight;
    float dw = diffuseWeight / sumWeight, sw = specularWeight / sumWeight;
    f3 wm = normalize(add(wo, wi));
    return dw * d_getPDFDiffuse(wi) + sw * d_getPDFSpecular(d, wm, wo);
}

/* ------------------------------------------------------------------ MetaMaterialGlass (BSDFs.h:404-479) */
static inline f3 metaglass_sample(int mm, f3 rho, float ior, f3 wo, f3* wi, float* pdf, cmj_state* st)
{
    float ior_o = 1.0f, ior_i = ior, sign = 1.0f;
    f3 lwo = wo, lwi = V1(0.0f);
    f3 n = V(0, 1, 0);
    if (wo.y < 0.0f) { ior_o = ior; ior_i = 1.0f; lwo.y = -lwo.y; sign = -1.0f; }
    const float fr = schlick_ior(mm, ior_o, ior_i, lwo, n);
    f3 evalbsdf;
    float p = cmj_1d(st);
    if (p < fr) {
        lwi = reflect3(neg(lwo), n);
        *pdf = 1; evalbsdf = divs(rho, fabsf(lwi.y));
    } else {
        f3 t;
        if (refract3(lwo, n, ior_o, ior_i, &t)) {
            lwi = reflect3(neg(t), V(0, -1, 0)); /* negative refractive index: tangential flip (:454) */
            *pdf = 1; evalbsdf = divs(rho, fabsf(lwi.y));
        } else {
            lwi = reflect3(neg(lwo), n);
            *pdf = 1; evalbsdf = divs(rho, fabsf(lwi.y));
        }
    }
    *wi = lwi;
    wi->y = sign * wi->y;
    return evalbsdf;
}

/* ------------------------------------------------------------------ EnagyConservationGGX (BSDFs.h:483-852) */
typedef struct { f3 F0; float alpha; int mm; } msggx;
static inline float ms_C1(float h) { return fminf(1.0f, fmaxf(0.0f, 0.5f * (h + 1.0f))); } /* :494-500 */
static inline float ms_invC1(float U) { return fmaxf(-1.0f, fminf(1.0f, 2.0f * U - 1.0f)); } /* :502-505 */
static inline float ms_Lambda(const msggx* g, f3 v) /* :525-532 */
{
    if (v.y > 0.9999f) return 0.0f;
    if (v.y < -0.9999f) return -1.0f;
    float a = g->alpha;
    float delta = 1.0f + (a * a * v.x * v.x + a * a * v.z * v.z) / (v.y * v.y);
    float sg = (v.y > 0.0f) ? 1.0f : -1.0f;
    return (float)((-1.0 + (double)(sg * sqrtf(delta))) / (double)2.0f);
}
static inline float ms_G1_Height(const msggx* g, f3 wi, float h0) /* :551-563 */
{
    if (wi.y > 0.9999f) return 1.0f;
    if (wi.y <= 0.0f) return 0.0f;
    const float C1_h0 = ms_C1(h0);
    const float Lambda = ms_Lambda(g, wi);
    return m_powf(g->mm, C1_h0, Lambda);
}
static inline float ms_sampleHeight(const msggx* g, f3 wr, float hr, float U) /* :566-586 */
{
    if (wr.y > 0.9999f) return FLT_MAX;
    if (wr.y < -0.9999f) return ms_invC1(U * ms_C1(hr));
    if (fabsf(wr.y) < 0.0001f) return hr;
    const float G_1_ = ms_G1_Height(g, wr, hr);
    if (U > 1.0f - G_1_) return FLT_MAX;
    return ms_invC1(ms_C1(hr) / m_powf(g->mm, (1.0f - U), 1.0f / ms_Lambda(g, wr)));
}
static inline f3 ms_samplePhase(const msggx* g, f3 wi, cmj_state* st, f3* weight) /* :737-746 */
{
    const f2 uv = cmj_2d(st);
    f3 wm = sample_visible_normal(g->mm, g->alpha, uv, wi);
    const f3 wo = add(neg(wi), muls(muls(wm, 2.0f), dot(wi, wm)));
    *weight = schlick3(g->mm, g->F0, wi, wm);
    return wo;
}
static inline f3 ms_sample(const msggx* g, f3 wi, f3* wo, int* order, cmj_state* st) /* :784-819 */
{
    f3 wr = neg(wi);
    float hr = 1.0f + ms_invC1(0.999f);
    *order = 0;
    f3 weight = V1(1.0f);
    for (;;) {
        float U = cmj_1d(st);
        hr = ms_sampleHeight(g, wr, hr, U);
        if (hr == FLT_MAX) break;
        else (*order)++;
        if (*order > 5) { *wo = V(0, 0, 1); return V(0, 0, 0); }
        f3 weight_1;
        wr = ms_samplePhase(g, neg(wr), st, &weight_1);
        weight = mul(weight, weight_1);
        if ((hr != hr) || (wr.z != wr.z)) return V(0, 0, 1);
    }
    *wo = wr;
    return weight;
}
static inline f3 msggx_sampleBSDF(const msggx* g, f3 wo, f3* wi, cmj_state* st, float* pdf) /* :843-851 */
{
    int order;
    f3 bsdf = ms_sample(g, wo, wi, &order, st);
    if (wi->y < 0.0f || order > 5) return V1(0.0f); /* pdf left as the caller initialised it */
    *pdf = fabsf(wi->y);
    return bsdf;
}

/* ------------------------------------------------------------------ BSDF dispatch (BSDFs.h:979-1038) */
typedef struct { int is_specular, is_ggx; float ior; disney dis; msggx eggx; int mm; } bsdf_t;
static inline void bsdf_init(bsdf_t* b, const payload* p, int mm, const uint8_t* lut, int lw, int lh)
{
    b->is_specular = p->is_specular;
    b->ior = p->ior;
    disney_init(&b->dis, p, mm, lut, lw, lh);
    b->eggx.F0 = p->basecolor;
    b->eggx.alpha = clampf(p->roughness * p->roughness, 0.0001f, 1.0f);
    b->eggx.mm = mm;
    b->is_ggx = p->metallic > 0.5f;
    b->mm = mm;
}
static inline f3 bsdf_eval(const bsdf_t* b, f3 wo, f3 wi)
{
    if (b->is_specular) return V1(0.0f);
    return disney_eval(&b->dis, wo, wi);
}
static inline f3 bsdf_sample(const bsdf_t* b, f3 wo, f3* wi, float* pdf, cmj_state* st)
{
    if (b->is_specular) return metaglass_sample(b->mm, V1(1.0f), b->ior, wo, wi, pdf, st);
    if (!b->is_ggx) return disney_sample(&b->dis, wo, wi, pdf, st);
    return msggx_sampleBSDF(&b->eggx, wo, wi, st, pdf);
}
static inline float bsdf_pdf(const bsdf_t* b, f3 wo, f3 wi)
{
    if (b->is_specular) return 0.0f;
    return disney_pdf(&b->dis, wo, wi);
}

static void payload_from_material(payload* p, const hjo_material* m)
{
    p->basecolor = V(m->basecolor[0], m->basecolor[1], m->basecolor[2]);
    p->metallic = m->metallic; p->roughness = m->roughness; p->sheen = m->sheen;
    p->clearcoat = m->clearcoat; p->ior = m->ior; p->transmission = m->transmission;
    p->is_specular = m->ideal_specular;
    p->emission = V(m->emission[0], m->emission[1], m->emission[2]);
    p->is_light = m->is_light; p->is_thinfilm = m->is_thinfilm;
}
void hjo_bsdf_sample(int mm, const hjo_material* mat, int which, const float* wo, uint32_t* s5,
                     float* f, float* wi, float* pdf)
{
    payload p; payload_init(&p); payload_from_material(&p, mat);
    bsdf_t b; bsdf_init(&b, &p, mm, 0, 0, 0);
    cmj_state st = state_from5(s5);
    f3 w = V(wi[0], wi[1], wi[2]), r;
    f3 o = V(wo[0], wo[1], wo[2]);
    if (which == 0) r = disney_sample(&b.dis, o, &w, pdf, &st);
    else if (which == 1) r = metaglass_sample(mm, V1(1.0f), b.ior, o, &w, pdf, &st);
    else if (which == 2) r = msggx_sampleBSDF(&b.eggx, o, &w, &st, pdf);
    else r = bsdf_sample(&b, o, &w, pdf, &st);
    s5[3] = st.depth;
    f[0] = r.x; f[1] = r.y; f[2] = r.z; wi[0] = w.x; wi[1] = w.y; wi[2] = w.z;
}
void hjo_bsdf_eval(int mm, const hjo_material* mat, const float* wo, const float* wi,
                   const uint8_t* lut, int lw, int lh, float* f)
{
    payload p; payload_init(&p); payload_from_material(&p, mat);
    bsdf_t b; bsdf_init(&b, &p, mm, lut, lw, lh);
    f3 r = bsdf_eval(&b, V(wo[0], wo[1], wo[2]), V(wi[0], wi[1], wi[2]));
    f[0] = r.x; f[1] = r.y; f[2] = r.z;
}
float hjo_bsdf_pdf(int mm, const hjo_material* mat, const float* wo, const float* wi)
{
    payload p; payload_init(&p); payload_from_material(&p, mat);
    bsdf_t b; bsdf_init(&b, &p, mm, 0, 0, 0);
    return bsdf_pdf(&b, V(wo[0], wo[1], wo[2]), V(wi[0], wi[1], wi[2]));
}

/* ------------------------------------------------------------------ world-space scene + BVH (build-defined:
 * replaces the closed OptiX GAS/IAS, renderer.h:319-490, and __closesthit__ch / __miss__ms) */
typedef struct { f3 v0, v1, v2; } wtri;
typedef struct { f3 n0, n1, n2; f2 t0, t1, t2; uint32_t mat, inst; } wshade;
typedef struct { float lo[3], hi[3]; uint32_t left, count; /* count>0: leaf [left,left+count) into order[] ; else children left,left+1 */ } bnode;

struct hjo_ctx {
    hjo_scene sc;
    int mm;
    wtri* tri; wshade* shade;     /* by global prim id */
    bnode* nodes; uint32_t n_nodes;
    uint32_t* order;              /* leaf order -> global prim id */
};

/* canonical ray/triangle test (Moeller-Trumbore, explicit fmaf).  Same sequence in the HIP kernel. */
static inline float dotf(f3 a, f3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
static inline f3 crossf(f3 a, f3 b)
{
    return V(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x)));
}
static inline int ray_tri(const wtri* T, f3 o, f3 d, float tmin, float tmax, float* t, float* b1, float* b2)
{
    f3 e1 = sub(T->v1, T->v0), e2 = sub(T->v2, T->v0);
    f3 p = crossf(d, e2);
    float det = dotf(e1, p);
    if (det == 0.0f) return 0;
    float inv = 1.0f / det;
    f3 tv = sub(o, T->v0);
    float u = dotf(tv, p) * inv;
    if (!(u >= 0.0f && u <= 1.0f)) return 0;
    f3 q = crossf(tv, e1);
    float v = dotf(d, q) * inv;
    if (!(v >= 0.0f && u + v <= 1.0f)) return 0;
    float tt = dotf(e2, q) * inv;
    if (!(tt > tmin && tt < tmax)) return 0;
    *t = tt; *b1 = u; *b2 = v;
    return 1;
}

static void build_world(hjo_ctx* c)
{
    const hjo_scene* s = &c->sc;
    for (uint32_t i = 0; i < s->n_instances; i++) {
        uint32_t t0 = s->prim_offsets[i];
        uint32_t t1 = (i + 1 < s->n_instances) ? s->prim_offsets[i + 1] : s->n_tris;
        const float* M = s->transforms + 12 * i;
        const float* Mi = s->inv_transforms + 12 * i;
        for (uint32_t t = t0; t < t1; t++) {
            f3 v[3], n[3]; f2 uv[3];
            for (int k = 0; k < 3; k++) {
                uint32_t idx = s->indices[3 * t + k];
                v[k] = transform_position(M, V(s->vertices[3 * idx], s->vertices[3 * idx + 1], s->vertices[3 * idx + 2]));
                n[k] = normalize(transform_normal(Mi, V(s->normals[3 * idx], s->normals[3 * idx + 1], s->normals[3 * idx + 2])));
                uv[k].x = s->texcoords[2 * idx]; uv[k].y = s->texcoords[2 * idx + 1];
            }
            c->tri[t].v0 = v[0]; c->tri[t].v1 = v[1]; c->tri[t].v2 = v[2];
            c->shade[t].n0 = n[0]; c->shade[t].n1 = n[1]; c->shade[t].n2 = n[2];
            c->shade[t].t0 = uv[0]; c->shade[t].t1 = uv[1]; c->shade[t].t2 = uv[2];
            c->shade[t].mat = s->material_ids[t]; c->shade[t].inst = i;
        }
    }
}

typedef struct { float lo[3], hi[3], c[3]; } tbox;
static uint32_t build_rec(hjo_ctx* c, tbox* tb, uint32_t first, uint32_t count, uint32_t* next)
{
    uint32_t me = (*next)++;
    bnode* nd = &c->nodes[me];
    float clo[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, chi[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
    for (int a = 0; a < 3; a++) { nd->lo[a] = FLT_MAX; nd->hi[a] = -FLT_MAX; }
    for (uint32_t i = first; i < first + count; i++) {
        const tbox* b = &tb[c->order[i]];
        for (int a = 0; a < 3; a++) {
            nd->lo[a] = fminf(nd->lo[a], b->lo[a]); nd->hi[a] = fmaxf(nd->hi[a], b->hi[a]);
            clo[a] = fminf(clo[a], b->c[a]); chi[a] = fmaxf(chi[a], b->c[a]);
        }
    }
    if (count <= 2) { nd->left = first; nd->count = count; return me; }
    int ax = 0; float ext = chi[0] - clo[0];
    for (int a = 1; a < 3; a++) if (chi[a] - clo[a] > ext) { ext = chi[a] - clo[a]; ax = a; }
    uint32_t mid;
    if (ext <= 0.0f) mid = first + count / 2;
    else {
        float sp = 0.5f * (clo[ax] + chi[ax]);
        uint32_t i = first, j = first + count;
        while (i < j) {
            if (tb[c->order[i]].c[ax] < sp) i++;
            else { j--; uint32_t tmp = c->order[i]; c->order[i] = c->order[j]; c->order[j] = tmp; }
        }
        mid = i;
        if (mid == first || mid == first + count) mid = first + count / 2;
    }
    nd->count = 0;
    /* depth-first allocation: the left child is always me+1; the right child index is recovered by
     * subtree_end() into a side table after the build (nodes[] may not be touched through nd after recursion). */
    uint32_t li = build_rec(c, tb, first, mid - first, next);
    (void)build_rec(c, tb, mid, first + count - mid, next);
    c->nodes[me].left = li;
    c->nodes[me].count = 0;
    return me;
}

static void build_bvh(hjo_ctx* c)
{
    uint32_t n = c->sc.n_tris;
    tbox* tb = (tbox*)malloc(sizeof(tbox) * (n ? n : 1));
    float smax = 0.0f;
    for (uint32_t t = 0; t < n; t++) {
        const wtri* T = &c->tri[t];
        const float vx[3] = { T->v0.x, T->v1.x, T->v2.x }, vy[3] = { T->v0.y, T->v1.y, T->v2.y }, vz[3] = { T->v0.z, T->v1.z, T->v2.z };
        const float* vv[3] = { vx, vy, vz };
        for (int a = 0; a < 3; a++) {
            tb[t].lo[a] = fminf(vv[a][0], fminf(vv[a][1], vv[a][2]));
            tb[t].hi[a] = fmaxf(vv[a][0], fmaxf(vv[a][1], vv[a][2]));
            tb[t].c[a] = 0.5f * (tb[t].lo[a] + tb[t].hi[a]);
            smax = fmaxf(smax, fmaxf(fabsf(tb[t].lo[a]), fabsf(tb[t].hi[a])));
        }
        c->order[t] = t;
    }
    /* conservative padding so the slab test can never cull a triangle the canonical test accepts */
    float pad = smax * (1.0f / 32768.0f);
    for (uint32_t t = 0; t < n; t++)
        for (int a = 0; a < 3; a++) { tb[t].lo[a] -= pad; tb[t].hi[a] += pad; }
    c->n_nodes = 0;
    if (n) { uint32_t next = 0; build_rec(c, tb, 0, n, &next); c->n_nodes = next; }
    free(tb);
}

/* With depth-first allocation the left child of node i is i+1; the right child is found by skipping the
 * left subtree.  We store the right child explicitly in a parallel array built after the fact. */
static uint32_t subtree_end(const hjo_ctx* c, uint32_t i)
{
    if (c->nodes[i].count) return i + 1;
    uint32_t e = subtree_end(c, i + 1);
    return subtree_end(c, e);
}

typedef struct { uint64_t box, tri; } tstat;

static inline int ray_box(const bnode* nd, f3 o, f3 inv, float tmin, float tmax)
{
    float t0 = (nd->lo[0] - o.x) * inv.x, t1 = (nd->hi[0] - o.x) * inv.x;
    float lo = fminf(t0, t1), hi = fmaxf(t0, t1);
    t0 = (nd->lo[1] - o.y) * inv.y; t1 = (nd->hi[1] - o.y) * inv.y;
    lo = fmaxf(lo, fminf(t0, t1)); hi = fminf(hi, fmaxf(t0, t1));
    t0 = (nd->lo[2] - o.z) * inv.z; t1 = (nd->hi[2] - o.z) * inv.z;
    lo = fmaxf(lo, fminf(t0, t1)); hi = fminf(hi, fmaxf(t0, t1));
    lo = fmaxf(lo, tmin); hi = fminf(hi, tmax);
    return lo <= hi * 1.0000004f;
}

/* closest hit with the order-independent tie rule: smaller t wins, equal t -> smaller global prim id */
static int trace_closest(const hjo_ctx* c, const uint32_t* right, f3 o, f3 d, float tmin, float tmax,
                         int use_bvh, float* ot, float* ob1, float* ob2, tstat* ts)
{
    int best = -1; float bt = tmax, bb1 = 0, bb2 = 0;
    if (!use_bvh) {
        for (uint32_t p = 0; p < c->sc.n_tris; p++) {
            float t, b1, b2;
            if (ray_tri(&c->tri[p], o, d, tmin, tmax, &t, &b1, &b2))
                if (best < 0 || t < bt || (t == bt && (int)p < best)) { best = (int)p; bt = t; bb1 = b1; bb2 = b2; }
        }
    } else if (c->n_nodes) {
        f3 inv = V(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
        uint32_t stack[128]; int sp = 0; stack[sp++] = 0;
        while (sp) {
            uint32_t i = stack[--sp];
            const bnode* nd = &c->nodes[i];
            if (ts) ts->box++;
            if (!ray_box(nd, o, inv, tmin, (best < 0) ? tmax : bt)) continue;
            if (nd->count) {
                for (uint32_t k = nd->left; k < nd->left + nd->count; k++) {
                    uint32_t p = c->order[k];
                    float t, b1, b2;
                    if (ts) ts->tri++;
                    if (ray_tri(&c->tri[p], o, d, tmin, tmax, &t, &b1, &b2))
                        if (best < 0 || t < bt || (t == bt && (int)p < best)) { best = (int)p; bt = t; bb1 = b1; bb2 = b2; }
                }
            } else {
                stack[sp++] = right[i];
                stack[sp++] = i + 1;
            }
        }
    }
    *ot = bt; *ob1 = bb1; *ob2 = bb2;
    return best;
}
static int trace_any(const hjo_ctx* c, const uint32_t* right, f3 o, f3 d, float tmin, float tmax, int use_bvh, tstat* ts)
{
    float t, b1, b2;
    if (!use_bvh) {
        for (uint32_t p = 0; p < c->sc.n_tris; p++)
            if (ray_tri(&c->tri[p], o, d, tmin, tmax, &t, &b1, &b2)) return 1;
        return 0;
    }
    if (!c->n_nodes) return 0;
    f3 inv = V(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    uint32_t stack[128]; int sp = 0; stack[sp++] = 0;
    while (sp) {
        uint32_t i = stack[--sp];
        const bnode* nd = &c->nodes[i];
        if (ts) ts->box++;
        if (!ray_box(nd, o, inv, tmin, tmax)) continue;
        if (nd->count) {
            for (uint32_t k = nd->left; k < nd->left + nd->count; k++) {
                if (ts) ts->tri++;
                if (ray_tri(&c->tri[c->order[k]], o, d, tmin, tmax, &t, &b1, &b2)) return 1;
            }
        } else { stack[sp++] = right[i]; stack[sp++] = i + 1; }
    }
    return 0;
}

/* ------------------------------------------------------------------ context */
static uint32_t* g_right_of(hjo_ctx* c);
struct hjo_ctx_ext { hjo_ctx base; uint32_t* right; };

hjo_ctx* hjo_create(const hjo_scene* scene, int math_mode)
{
    struct hjo_ctx_ext* e = (struct hjo_ctx_ext*)calloc(1, sizeof(*e));
    hjo_ctx* c = &e->base;
    c->sc = *scene; c->mm = math_mode;
    uint32_t n = scene->n_tris;
    c->tri = (wtri*)calloc(n ? n : 1, sizeof(wtri));
    c->shade = (wshade*)calloc(n ? n : 1, sizeof(wshade));
    c->nodes = (bnode*)calloc(2 * (n ? n : 1), sizeof(bnode));
    c->order = (uint32_t*)calloc(n ? n : 1, sizeof(uint32_t));
    build_world(c);
    build_bvh(c);
    e->right = (uint32_t*)calloc(c->n_nodes ? c->n_nodes : 1, sizeof(uint32_t));
    for (uint32_t i = 0; i < c->n_nodes; i++)
        if (!c->nodes[i].count) e->right[i] = subtree_end(c, i + 1);
    return c;
}
static uint32_t* g_right_of(hjo_ctx* c) { return ((struct hjo_ctx_ext*)c)->right; }
void hjo_destroy(hjo_ctx* c)
{
    if (!c) return;
    free(g_right_of(c)); free(c->tri); free(c->shade); free(c->nodes); free(c->order); free(c);
}
int hjo_trace_closest(hjo_ctx* c, const float* o, const float* d, float tmin, float tmax, int use_bvh, float* out)
{
    return trace_closest(c, g_right_of(c), V(o[0], o[1], o[2]), V(d[0], d[1], d[2]), tmin, tmax, use_bvh, &out[0], &out[1], &out[2], 0);
}
int hjo_trace_any(hjo_ctx* c, const float* o, const float* d, float tmin, float tmax, int use_bvh)
{
    return trace_any(c, g_right_of(c), V(o[0], o[1], o[2]), V(d[0], d[1], d[2]), tmin, tmax, use_bvh, 0);
}

/* ------------------------------------------------------------------ per-thread tracing context */
typedef struct {
    hjo_ctx* c; const hjo_params* P; const uint32_t* right;
    hjo_stats st;
} tctx;

/* RayTrace (rt.h:43-69) + __closesthit__ch / __miss__ms (build-defined) */
static void RayTrace(tctx* T, f3 o, f3 d, float tmin, float tmax, payload* prd)
{
    hjo_ctx* c = T->c;
    payload_init(prd);
    float t, b1, b2; tstat ts = { 0, 0 };
    int prim = trace_closest(c, T->right, o, d, tmin, tmax, 1, &t, &b1, &b2, &ts);
    T->st.closest_rays++; T->st.box_tests_closest += ts.box; T->st.tri_tests_closest += ts.tri;
    if (prim < 0) { /* miss: constant sky (renderer.h:802-851 with use_IBL=false) */
        prd->is_hit = 0;
        if (c->sc.sky_rgba && c->sc.sky_w > 0 && c->sc.sky_h > 0) { /* tex2D(ibl_texture, u, v) * ibl_intensity */
            float dir[3] = { d.x, d.y, d.z }, e[3];
            hjo_sky_fetch(c->mm, c->sc.sky_rgba, c->sc.sky_w, c->sc.sky_h, dir, e);
            prd->emission = muls(V(e[0], e[1], e[2]), T->P->ibl_intensity);
        } else prd->emission = muls(V(T->P->sky[0], T->P->sky[1], T->P->sky[2]), T->P->ibl_intensity);
        return;
    }
    const wtri* W = &c->tri[prim]; const wshade* S = &c->shade[prim];
    float w0 = 1.0f - b1 - b2;
    prd->is_hit = 1;
    prd->position = add(add(muls(W->v0, w0), muls(W->v1, b1)), muls(W->v2, b2));
    prd->normal = add(add(muls(S->n0, w0), muls(S->n1, b1)), muls(S->n2, b2));
    prd->texcoord.x = S->t0.x * w0 + S->t1.x * b1 + S->t2.x * b2;
    prd->texcoord.y = S->t0.y * w0 + S->t1.y * b1 + S->t2.y * b2;
    payload_from_material(prd, &c->sc.materials[S->mat]);
    { /* material textures: factor x texel (glTF 2.0 semantics; build-defined, the closest-hit source is missing) */
        const hjo_material* m = &c->sc.materials[S->mat];
        if (m->basecolor_tex >= 0 && (uint32_t)m->basecolor_tex < c->sc.n_textures) {
            const hjo_texture* tx = &c->sc.textures[m->basecolor_tex];
            float e[3]; hjo_tex_fetch(tx->rgba8, (int)tx->width, (int)tx->height, tx->srgb, prd->texcoord.x, prd->texcoord.y, e);
            prd->basecolor = mul(prd->basecolor, V(e[0], e[1], e[2]));
        }
        if (m->metallic_roughness_tex >= 0 && (uint32_t)m->metallic_roughness_tex < c->sc.n_textures) {
            const hjo_texture* tx = &c->sc.textures[m->metallic_roughness_tex];
            float e[3]; hjo_tex_fetch(tx->rgba8, (int)tx->width, (int)tx->height, tx->srgb, prd->texcoord.x, prd->texcoord.y, e);
            prd->roughness = prd->roughness * e[1];
            prd->metallic = prd->metallic * e[2];
        }
        /* normal map (Material.normal_tex, gltfloader.h:1168-1175, bound at renderer.h:680; build-defined: glTF 2.0 tangent space).
         * Per-triangle tangent / bitangent from the world-space edges and uv deltas, tangent orthogonalised against the normalised
         * shading normal, bitangent = cross(N, T) with the sign of the geometric bitangent; n = normalize(T nx + B ny + N nz). */
        if (m->normal_tex >= 0 && (uint32_t)m->normal_tex < c->sc.n_textures) {
            const hjo_texture* tx = &c->sc.textures[m->normal_tex];
            f3 e1 = sub(W->v1, W->v0), e2 = sub(W->v2, W->v0);
            float du1 = S->t1.x - S->t0.x, dv1 = S->t1.y - S->t0.y, du2 = S->t2.x - S->t0.x, dv2 = S->t2.y - S->t0.y;
            float det = du1 * dv2 - du2 * dv1;
            if (det != 0.0f) {
                float r = 1.0f / det;
                f3 tg = muls(sub(muls(e1, dv2), muls(e2, dv1)), r), bt = muls(sub(muls(e2, du1), muls(e1, du2)), r);
                f3 ns = normalize(prd->normal);
                f3 tp = normalize(sub(tg, muls(ns, dot(ns, tg))));
                f3 bp = cross(ns, tp);
                if (dot(bp, bt) < 0.0f) bp = neg(bp);
                float e[3]; hjo_tex_fetch(tx->rgba8, (int)tx->width, (int)tx->height, tx->srgb, prd->texcoord.x, prd->texcoord.y, e);
                f3 nm = V(2.0f * e[0] - 1.0f, 2.0f * e[1] - 1.0f, 2.0f * e[2] - 1.0f);
                f3 mapped = add(add(muls(tp, nm.x), muls(bp, nm.y)), muls(ns, nm.z));
                float l2 = dot(mapped, mapped);
                if (l2 > 0.0f && l2 - l2 == 0.0f) prd->normal = muls(mapped, 1.0f / sqrtf(l2));
            }
        }
    }
    prd->primitive_id = prim; prd->instance_id = (int)S->inst;
    T->st.shaded_hits++;
}
/* TraceOcculution (rt.h:15-41) + __closesthit__shadow */
static int TraceOcclusion(tctx* T, f3 o, f3 d, float tmin, float tmax)
{
    tstat ts = { 0, 0 };
    int h = trace_any(T->c, T->right, o, d, tmin, tmax, 1, &ts);
    T->st.shadow_rays++; T->st.box_tests_shadow += ts.box; T->st.tri_tests_shadow += ts.tri;
    return h;
}

/* light_sample (light_sample.h:9-75) */
static f3 light_sample(tctx* T, cmj_state* st, float* pdf, f3* normal, f3* emission, int* valid)
{
    const hjo_scene* s = &T->c->sc;
    if (s->n_lights < 1) { *pdf = -1.0f; *normal = V1(0.0f); *emission = V1(0.0f); *valid = 0; return V1(0.0f); }
    *valid = 1;
    float p = cmj_1d(st);
    int index = (int)(p * s->n_lights);
    if (index == (int)s->n_lights) index--;
    uint32_t prim_index = s->light_prim_ids[index];
    uint32_t left = 0U, right = s->n_instances - 1, middle = (left + right) / 2U;
    while (left <= right) {
        if (s->prim_offsets[middle] <= prim_index) left = middle + 1;
        else right = middle - 1;
        middle = (left + right) / 2;
    }
    float select_pdf = 1.0f / s->n_lights;
    uint32_t inst = middle;
    const float* M = s->transforms + 12 * inst; const float* Mi = s->inv_transforms + 12 * inst;
    uint32_t i0 = s->indices[prim_index * 3 + 0], i1 = s->indices[prim_index * 3 + 1], i2 = s->indices[prim_index * 3 + 2];
#define VTX(a, i) V(a[3 * (i)], a[3 * (i) + 1], a[3 * (i) + 2])
    const f3 v0 = transform_position(M, VTX(s->vertices, i0));
    const f3 v1 = transform_position(M, VTX(s->vertices, i1));
    const f3 v2 = transform_position(M, VTX(s->vertices, i2));
    const f3 n0 = transform_normal(Mi, VTX(s->normals, i0));
    const f3 n1 = transform_normal(Mi, VTX(s->normals, i1));
    const f3 n2 = transform_normal(Mi, VTX(s->normals, i2));
    const float light_area = length3(cross(sub(v1, v0), sub(v2, v0))) * 0.5f;
    f2 xi = cmj_2d(st);
    float f1, f2_, f3_; /* `sqrt(xi.x)`: un-suffixed in the reference (light_sample.h:62-64) */
    if (T->c->mm == 2) {
        double sq = sqrt((double)xi.x);
        f1 = (float)(1.0 - sq); f2_ = (float)(sq * (double)(1.0f - xi.y)); f3_ = (float)(sq * (double)xi.y);
    } else {
        f1 = 1.0f - sqrtf(xi.x); f2_ = sqrtf(xi.x) * (1.0f - xi.y); f3_ = sqrtf(xi.x) * xi.y;
    }
    const f3 light_position = add(add(muls(v0, f1), muls(v1, f2_)), muls(v2, f3_));
    const f3 light_normal = normalize(add(add(muls(n0, f1), muls(n1, f2_)), muls(n2, f3_)));
    float pd = (float)(1.0 / (double)light_area);
    pd *= select_pdf;
    *pdf = pd; *normal = light_normal;
    *emission = V(s->light_prim_emission[3 * index], s->light_prim_emission[3 * index + 1], s->light_prim_emission[3 * index + 2]);
    T->st.light_samples++;
    return light_position;
}
/* getLightPDF (light_sample.h:77-92) */
static float getLightPDF(tctx* T, uint32_t primID, uint32_t instID)
{
    const hjo_scene* s = &T->c->sc;
    uint32_t i0 = s->indices[primID * 3 + 0], i1 = s->indices[primID * 3 + 1], i2 = s->indices[primID * 3 + 2];
    const float* M = s->transforms + 12 * instID;
    const f3 v0 = transform_position(M, VTX(s->vertices, i0));
    const f3 v1 = transform_position(M, VTX(s->vertices, i1));
    const f3 v2 = transform_position(M, VTX(s->vertices, i2));
    float light_area = length3(cross(sub(v1, v0), sub(v2, v0))) * 0.5f;
    return 1.0f / (light_area * s->n_lights);
}

/* NEE (rt.h:162-281) */
/* Test-infrastructure aid: with hjo_set_trace(1) the NEE integrator narrates a sample on stderr (used once to find the expression behind
 * the NaN samples of the C2 frame). */
static int hjo_trace = 0;
void hjo_set_trace(int on) { hjo_trace = on; }
#define TR(...) do { if (hjo_trace) fprintf(stderr, __VA_ARGS__); } while (0)
static f3 NEE(tctx* T, f3 o, f3 d, cmj_state state, f3* aov_albedo, f3* aov_normal)
{
    const hjo_scene* s = &T->c->sc; int mm = T->c->mm;
    f3 LTE = V1(0.0f), throughput = V1(1.0f);
    f3 ro = o, rd = d;
    for (int depth = 0; depth < 10; depth++) {
        float russian_p = fmaxf(throughput.x, fmaxf(throughput.y, throughput.z));
        if (russian_p < cmj_1d(&state)) break;
        throughput = divs(throughput, russian_p);
        payload prd;
        RayTrace(T, ro, rd, 0.001f, 1e16f, &prd);
        if (depth == 0) { *aov_albedo = prd.basecolor; *aov_normal = prd.normal; }
        if (!prd.is_hit) { if (depth == 0) LTE = add(LTE, mul(throughput, prd.emission)); break; }
        if (prd.is_light) { if (depth == 0) LTE = add(LTE, mul(throughput, prd.emission)); break; }
        bsdf_t bs; bsdf_init(&bs, &prd, mm, s->lut_rgba, s->lut_w, s->lut_h);
        f3 t, b, n = prd.normal;
        orthonormal_basis(n, &t, &b);
        f3 local_wo = world_to_local(neg(rd), t, n, b);
        TR("depth %d: prim %d basecolor %g %g %g metallic %g roughness %g sheen %g clearcoat %g ior %g specular %d thinfilm %d | n %g %g %g (|n|^2 %.9g) wo_local %g %g %g thr %g %g %g\n", depth, (int)prd.primitive_id,
           prd.basecolor.x, prd.basecolor.y, prd.basecolor.z, prd.metallic, prd.roughness, prd.sheen, prd.clearcoat, prd.ior, (int)prd.is_specular, (int)prd.is_thinfilm,
           n.x, n.y, n.z, dot(n, n), local_wo.x, local_wo.y, local_wo.z, throughput.x, throughput.y, throughput.z);
        { /* NEE */
            float light_pdf; f3 light_color, light_normal; int valid;
            f3 light_position = light_sample(T, &state, &light_pdf, &light_normal, &light_color, &valid);
            if (valid) { /* light_prim_count < 1 leaves emission uninitialised in the reference: defined as no contribution */
                f3 so = prd.position;
                f3 sd = normalize(sub(light_position, so));
                float light_distance = length3(sub(light_position, so));
                float ipsiron_distance = 0.001f;
                /* The contribution does not depend on the shadow ray; an exactly-zero one (glass: evaluateBSDF == 0) cannot
                 * change LTE (x + 0 == x), so its shadow ray is not traced (and not counted).  Same values as the reference's
                 * trace-then-evaluate order, rt.h:236-259. */
                float cosine1 = absdot(n, sd);
                float cosine2 = absdot(light_normal, neg(sd));
                f3 local_wi = world_to_local(sd, t, n, b);
                f3 bsdf = bsdf_eval(&bs, local_wo, local_wi);
                float G = cosine2 / (light_distance * light_distance);
                f3 contrib = mul(mul(throughput, divs(muls(muls(bsdf, G), cosine1), light_pdf)), light_color);
                TR("   nee: wi_local %g %g %g bsdf_eval %g %g %g G %g cos1 %g light_pdf %g contrib %g %g %g\n", local_wi.x, local_wi.y, local_wi.z, bsdf.x, bsdf.y, bsdf.z, G, cosine1, light_pdf, contrib.x, contrib.y, contrib.z);
                if (!(contrib.x == 0.0f && contrib.y == 0.0f && contrib.z == 0.0f) &&
                    !TraceOcclusion(T, so, sd, 0.001f, light_distance - ipsiron_distance))
                    LTE = add(LTE, contrib);
            }
        }
        float pdf = 1.0f;
        f3 local_wi = V(0.0f, 1.0f, 0.0f);
        (void)cmj_2d(&state); /* the reference draws and discards one sample here (rt.h:266) */
        f3 bsdf = bsdf_sample(&bs, local_wo, &local_wi, &pdf, &state);
        f3 wi = local_to_world(local_wi, t, n, b);
        throughput = mul(throughput, divs(muls(bsdf, fabsf(dot(wi, n))), pdf));
        TR("   sample: wi_local %g %g %g bsdf %g %g %g pdf %g -> thr %g %g %g  LTE %g %g %g\n", local_wi.x, local_wi.y, local_wi.z, bsdf.x, bsdf.y, bsdf.z, pdf, throughput.x, throughput.y, throughput.z, LTE.x, LTE.y, LTE.z);
        ro = prd.position; rd = wi;
    }
    return LTE;
}
/* Pathtrace (rt.h:85-159) */
static f3 Pathtrace(tctx* T, f3 o, f3 d, cmj_state state, f3* aov_albedo, f3* aov_normal)
{
    const hjo_scene* s = &T->c->sc; int mm = T->c->mm;
    f3 LTE = V1(0.0f), throughput = V1(1.0f);
    f3 ro = o, rd = d;
    for (int depth = 0; depth < 10; depth++) {
        float russian_p = fmaxf(throughput.x, fmaxf(throughput.y, throughput.z));
        if (russian_p < cmj_1d(&state)) break;
        throughput = divs(throughput, russian_p);
        payload prd;
        RayTrace(T, ro, rd, 0.001f, 1e16f, &prd);
        if (depth == 0) { *aov_albedo = prd.basecolor; *aov_normal = prd.normal; }
        if (!prd.is_hit) { LTE = add(LTE, mul(throughput, prd.emission)); break; }
        if (prd.is_light) { LTE = add(LTE, mul(throughput, prd.emission)); break; }
        bsdf_t bs; bsdf_init(&bs, &prd, mm, s->lut_rgba, s->lut_w, s->lut_h);
        f3 t, b, n = prd.normal;
        orthonormal_basis(n, &t, &b);
        float pdf = 1.0f;
        f3 local_wo = world_to_local(neg(rd), t, n, b);
        f3 local_wi = V(0.0f, 1.0f, 0.0f);
        f3 bsdf = bsdf_sample(&bs, local_wo, &local_wi, &pdf, &state);
        f3 wi = local_to_world(local_wi, t, n, b);
        throughput = mul(throughput, divs(muls(bsdf, fabsf(dot(wi, n))), pdf));
        ro = prd.position; rd = wi;
    }
    return LTE;
}
/* MIS (rt.h:284-440).  pt_pdf is uninitialised in the reference when msGGX returns early; defined as 1. */
static f3 MIS(tctx* T, f3 o, f3 d, cmj_state* statep, f3* aov_albedo, f3* aov_normal)
{
    const hjo_scene* s = &T->c->sc; int mm = T->c->mm;
    cmj_state state = *statep;
    f3 LTE = V1(0.0f), throughput = V1(1.0f);
    f3 ro = o, rd = d;
    for (int depth = 0; depth < 10; depth++) {
        float russian_p = fmaxf(throughput.x, fmaxf(throughput.y, throughput.z));
        if (russian_p < cmj_1d(&state)) break;
        throughput = divs(throughput, russian_p);
        payload prd;
        RayTrace(T, ro, rd, 0.001f, 1e16f, &prd);
        if (depth == 0) { *aov_albedo = prd.basecolor; *aov_normal = prd.normal; }
        if (!prd.is_hit) { if (depth == 0) LTE = add(LTE, mul(throughput, prd.emission)); break; }
        if (prd.is_light) { if (depth == 0) LTE = add(LTE, mul(throughput, prd.emission)); break; }
        bsdf_t bs; bsdf_init(&bs, &prd, mm, s->lut_rgba, s->lut_w, s->lut_h);
        f3 t, b, n = prd.normal;
        orthonormal_basis(n, &t, &b);
        f3 local_wo = world_to_local(neg(rd), t, n, b);
        { /* NEE */
            float light_pdf; f3 light_normal, light_emission; int valid;
            f3 light_position = light_sample(T, &state, &light_pdf, &light_normal, &light_emission, &valid);
            if (valid) {
                f3 light_direction = sub(light_position, prd.position);
                float light_distance = length3(light_direction);
                light_direction = normalize(light_direction);
                float cosine1 = absdot(n, light_direction);
                float cosine2 = absdot(light_normal, neg(light_direction));
                f3 local_wi = world_to_local(light_direction, t, n, b);
                f3 bsdf = bsdf_eval(&bs, local_wo, local_wi);
                float G = cosine2 / (light_distance * light_distance);
                float pt_pdf = bsdf_pdf(&bs, local_wo, local_wi) * G;
                float mis_weight = light_pdf / (light_pdf + pt_pdf);
                f3 contrib = mul(muls(mul(throughput, divs(muls(muls(bsdf, G), cosine1), light_pdf)), mis_weight), light_emission);
                if (!(contrib.x == 0.0f && contrib.y == 0.0f && contrib.z == 0.0f) &&
                    !TraceOcclusion(T, prd.position, light_direction, 0.001f, light_distance - 0.001f))
                    LTE = add(LTE, contrib);
            }
        }
        { /* Pathtrace */
            float pt_pdf = 1.0f; f3 local_wi = V(0.0f, 1.0f, 0.0f);
            f3 brdf = bsdf_sample(&bs, local_wo, &local_wi, &pt_pdf, &state);
            f3 wi = local_to_world(local_wi, t, n, b);
            float cosine1 = absdot(wi, n);
            payload lh;
            RayTrace(T, prd.position, wi, 0.001f, 1e16f, &lh);
            if (lh.is_hit) {
                if (lh.is_light) {
                    float cosine2 = absdot(neg(wi), lh.normal);
                    float light_distance = length3(sub(lh.position, prd.position));
                    float invG = light_distance * light_distance / cosine2;
                    float lightPdf = prd.is_specular ? 0.0f : getLightPDF(T, (uint32_t)lh.primitive_id, (uint32_t)lh.instance_id) * invG;
                    float mis_weight = pt_pdf / (pt_pdf + lightPdf);
                    LTE = add(LTE, divs(mul(mul(muls(muls(throughput, mis_weight), cosine1), lh.emission), brdf), pt_pdf));
                }
            } else {
                LTE = add(LTE, divs(mul(muls(mul(throughput, brdf), cosine1), lh.emission), pt_pdf));
            }
        }
        float pdf = 1.0f;
        f3 local_wi = V(0.0f, 1.0f, 0.0f);
        (void)cmj_2d(&state);
        f3 bsdf = bsdf_sample(&bs, local_wo, &local_wi, &pdf, &state);
        f3 wi = local_to_world(local_wi, t, n, b);
        throughput = mul(throughput, divs(muls(bsdf, fabsf(dot(wi, n))), pdf));
        ro = prd.position; rd = wi;
    }
    *statep = state;
    return LTE;
}

/* __raygen__rg body for one (pixel, sample) — build-defined (stale ptx:33-106) */
static void sample_one(tctx* T, uint32_t x, uint32_t y, uint32_t s, f3* L, f3* A, f3* N)
{
    const hjo_params* P = T->P;
    cmj_state st;
    st.n_spp = (uint64_t)P->frame * (uint64_t)P->spp + (uint64_t)s;
    st.scramble = P->seed; st.depth = 0; st.image_idx = x + y * P->width;
    f2 j = cmj_2d(&st);
    float W = (float)P->width, H = (float)P->height;
    float u = (2.0f * ((float)x + j.x) - W) / H;
    float v = (2.0f * ((float)y + j.y) - H) / H;
    f3 cp = V(P->cam_pos[0], P->cam_pos[1], P->cam_pos[2]);
    f3 cd = V(P->cam_dir[0], P->cam_dir[1], P->cam_dir[2]);
    f3 cu = V(P->cam_up[0], P->cam_up[1], P->cam_up[2]);
    f3 cr = V(P->cam_right[0], P->cam_right[1], P->cam_right[2]);
    f3 d = normalize(add(add(muls(cd, P->cam_f), muls(cr, u)), muls(cu, v)));
    *A = V1(0.0f); *N = V1(0.0f);
    if (P->integrator == HJO_INTEGRATOR_PT) *L = Pathtrace(T, cp, d, st, A, N);
    else if (P->integrator == HJO_INTEGRATOR_MIS) *L = MIS(T, cp, d, &st, A, N);
    else *L = NEE(T, cp, d, st, A, N);
    T->st.samples++;
    float sum = L->x + L->y + L->z;
    if (!(sum - sum == 0.0f)) { *L = V1(0.0f); T->st.nan_samples++; } /* NaN/Inf guard (build-defined) */
}

int hjo_sample(hjo_ctx* c, const hjo_params* P, uint32_t x, uint32_t y, uint32_t s, float* rad, float* alb, float* nor)
{
    tctx T; memset(&T, 0, sizeof(T)); T.c = c; T.P = P; T.right = g_right_of(c);
    f3 L, A, N; sample_one(&T, x, y, s, &L, &A, &N);
    rad[0] = L.x; rad[1] = L.y; rad[2] = L.z;
    if (alb) { alb[0] = A.x; alb[1] = A.y; alb[2] = A.z; }
    if (nor) { nor[0] = N.x; nor[1] = N.y; nor[2] = N.z; }
    return (int)T.st.nan_samples; /* 1: the sample was NaN / Inf and has been zeroed by the guard */
}

typedef struct { tctx T; float *color, *albedo, *normal; volatile uint32_t* next_row; uint32_t x0, y0, x1, y1; } job;

static void render_pixel(job* J, uint32_t x, uint32_t y)
{
    const hjo_params* P = J->T.P;
    /* build-defined accumulation order: samples are summed in order inside runs of chunk_spp
     * consecutive samples (a multiple of 8, at most 64 runs per pixel), and the run sums are added in run order.
     * With one run this is the plain in-order sum. */
    uint32_t n8 = (P->spp + 7u) / 8u;
    uint32_t chunk = 8u * ((n8 + 63u) / 64u);
    f3 sL = V1(0.0f), sA = V1(0.0f), sN = V1(0.0f);
    for (uint32_t s0 = 0; s0 < P->spp; s0 += chunk) {
        uint32_t s1 = s0 + chunk < P->spp ? s0 + chunk : P->spp;
        f3 cL = V1(0.0f), cA = V1(0.0f), cN = V1(0.0f);
        for (uint32_t s = s0; s < s1; s++) {
            f3 L, A, N; sample_one(&J->T, x, y, s, &L, &A, &N);
            cL = add(cL, L); cA = add(cA, A); cN = add(cN, N);
        }
        if (s0 == 0) { sL = cL; sA = cA; sN = cN; }
        else { sL = add(sL, cL); sA = add(sA, cA); sN = add(sN, cN); }
    }
    float inv = 1.0f / (float)P->spp;
    size_t pix = (size_t)x + (size_t)y * P->width;
    if (J->color) { float* o = J->color + 4 * pix; o[0] = sL.x * inv; o[1] = sL.y * inv; o[2] = sL.z * inv; o[3] = 1.0f; }
    if (J->albedo) { float* o = J->albedo + 4 * pix; o[0] = sA.x * inv; o[1] = sA.y * inv; o[2] = sA.z * inv; o[3] = 1.0f; }
    if (J->normal) { float* o = J->normal + 4 * pix; o[0] = sN.x * inv; o[1] = sN.y * inv; o[2] = sN.z * inv; o[3] = 1.0f; }
}
static void* worker(void* arg)
{
    job* J = (job*)arg;
    for (;;) {
        uint32_t y = __sync_fetch_and_add(J->next_row, 1);
        if (y >= J->y1) break;
        for (uint32_t x = J->x0; x < J->x1; x++) render_pixel(J, x, y);
    }
    return 0;
}
int hjo_render(hjo_ctx* c, const hjo_params* P, float* color, float* albedo, float* normal, int nthreads, hjo_stats* stats)
{
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    uint32_t x0 = P->x0, y0 = P->y0, x1 = P->x1, y1 = P->y1;
    if (x1 == 0) { x0 = 0; y0 = 0; x1 = P->width; y1 = P->height; }
    volatile uint32_t next = y0;
    job* jobs = (job*)calloc((size_t)nthreads, sizeof(job));
    pthread_t* th = (pthread_t*)calloc((size_t)nthreads, sizeof(pthread_t));
    for (int i = 0; i < nthreads; i++) {
        jobs[i].T.c = c; jobs[i].T.P = P; jobs[i].T.right = g_right_of(c);
        jobs[i].color = color; jobs[i].albedo = albedo; jobs[i].normal = normal;
        jobs[i].next_row = &next; jobs[i].x0 = x0; jobs[i].y0 = y0; jobs[i].x1 = x1; jobs[i].y1 = y1;
    }
    if (nthreads == 1) worker(&jobs[0]);
    else {
        for (int i = 0; i < nthreads; i++) pthread_create(&th[i], 0, worker, &jobs[i]);
        for (int i = 0; i < nthreads; i++) pthread_join(th[i], 0);
    }
    if (stats) {
        memset(stats, 0, sizeof(*stats));
        for (int i = 0; i < nthreads; i++) {
            const uint64_t* a = (const uint64_t*)&jobs[i].T.st; uint64_t* b = (uint64_t*)stats;
            for (size_t k = 0; k < sizeof(hjo_stats) / 8; k++) b[k] += a[k];
        }
    }
    free(jobs); free(th);
    return 0;
}

/* ------------------------------------------------------------------ output stage (renderer.h:73-101) */
void hjo_float4_to_srgb8(const float* rgba, uint8_t* out, uint32_t n)
{
    const float invGamma = 1.0f / 2.4f;
    for (uint32_t i = 0; i < n; i++) {
        for (int c = 0; c < 3; c++) {
            float col = rgba[4 * i + c];
            float powed = powf(col, invGamma);
            float sr = col < 0.0031308f ? 12.92f * col : 1.055f * powed - 0.055f;
            float q = sr * 256.0f;
            uint32_t u = (q > 0.0f) ? ((q >= 4294967040.0f) ? 4294967040u : (uint32_t)q) : 0u; /* negative/NaN is UB in the reference: 0 */
            out[4 * i + c] = (uint8_t)(u < 255u ? u : 255u);
        }
        out[4 * i + 3] = 255;
    }
}

/* kernel/color.h:10-29.  The double literals (1.0, 0.0) promote the surrounding sub-expressions exactly as written there. */
static float tonemap_uchimura1(float x, float P, float a, float m, float l, float c, float b)
{
    float l0 = ((P - m) * l) / a;
    float S0 = m + l0;
    float S1 = m + a * l0;
    float C2 = (a * P) / (P - S1);
    float CP = -C2 / P;
    /* smoothstep(0.0, m, x) (math.h:113-116), step(m + l0, x) (math.h:118-120) */
    float sx = fmaxf(0.0f, fminf((x - 0.0f) / (m - 0.0f), 1.0f));
    float w0 = (float)(1.0 - (double)(sx * sx * (3.0f - 2.0f * sx)));
    float w2 = (float)((m + l0) < x);
    float w1 = (float)(1.0 - (double)w0 - (double)w2);
    float T = (float)((double)m * pow((double)(x / m), (double)c) + (double)b);
    float S = (float)((double)P - (double)(P - S1) * exp((double)(CP * (x - S0))));
    float L = m + a * (x - m);
    return T * w0 + L * w1 + S * w2;
}
static float tonemap_uchimura(float x) { return tonemap_uchimura1(x, 1.0f, 1.0f, 0.22f, 0.4f, 1.33f, 0.0f); } /* color.h:31-39 */
static float tonemap_aces(float x) /* color.h:55-63 */
{
    const float a = 2.51f, b = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;
    return fmaxf(0.0f, fminf((x * (a * x + b)) / (x * (c * x + d) + e), 1.0f));
}

void hjo_tonemap_to_srgb8(const float* rgba, uint8_t* out, uint32_t n, int mode)
{
    float px[4];
    for (uint32_t i = 0; i < n; i++) {
        for (int c = 0; c < 3; c++) {
            float v = rgba[4 * (size_t)i + c];
            px[c] = mode == 1 ? tonemap_uchimura(v) : (mode == 2 ? tonemap_aces(v) : v);
        }
        px[3] = 1.0f;
        hjo_float4_to_srgb8(px, out + 4 * (size_t)i, 1);
    }
}
float hjo_tonemap(float x, int mode) { return mode == 1 ? tonemap_uchimura(x) : (mode == 2 ? tonemap_aces(x) : x); }

/* ------------------------------------------------------------------ denoise-mode replacement (build-defined; checker of
 * henjou-renderer_amd/csrc/hjr_denoise.hip.h, whose header states the algorithm: 5-pass edge-avoiding a-trous filter guided by
 * the albedo / normal AOVs, optional 2x bilinear upscale).  Data flow of renderer/denoiser.h:42-189 + renderer.h:1093-1120,
 * 1258-1270; the OptiX network itself is closed and is not reproduced.  mode: 0 Default (copy), 1 Denoise, 2 DenoiseUpScale2X. */
static inline float dn_weight(const float* a, const float* b, float rel, float phi)
{
    float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
    float d2 = dx * dx + dy * dy + dz * dz;
    float e = -(d2 / rel) / phi;
    e = fmaxf(e, -87.0f);
    return fminf(p_exp(e), 1.0f);
}
static void dn_atrous_pass(const float* in, const float* normal, const float* albedo, float* out, int W, int H, int step, float c_phi)
{
    static const float hk[5] = { 0.0625f, 0.25f, 0.375f, 0.25f, 0.0625f };
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const size_t c = ((size_t)y * W + x) * 4;
            const float s0 = (in[c] + in[c + 1]) + in[c + 2];
            const float rel = 0.01f + s0 * s0;
            float sx = 0.0f, sy = 0.0f, sz = 0.0f, cum = 0.0f;
            for (int j = -2; j <= 2; j++) {
                int yy = y + j * step; yy = yy < 0 ? 0 : (yy > H - 1 ? H - 1 : yy);
                for (int i = -2; i <= 2; i++) {
                    int xx = x + i * step; xx = xx < 0 ? 0 : (xx > W - 1 ? W - 1 : xx);
                    const size_t t = ((size_t)yy * W + xx) * 4;
                    const float wc = c_phi > 0.0f ? dn_weight(in + c, in + t, rel, c_phi) : 1.0f;
                    const float wn = dn_weight(normal + c, normal + t, 1.0f, 0.25f);
                    const float wa = dn_weight(albedo + c, albedo + t, 1.0f, 0.05f);
                    const float w = ((wc * wn) * wa) * (hk[j + 2] * hk[i + 2]);
                    sx = sx + in[t] * w; sy = sy + in[t + 1] * w; sz = sz + in[t + 2] * w;
                    cum = cum + w;
                }
            }
            out[c] = sx / cum; out[c + 1] = sy / cum; out[c + 2] = sz / cum; out[c + 3] = in[c + 3];
        }
}
int hjo_denoise(int mode, uint32_t in_w, uint32_t in_h, const float* color, const float* albedo, const float* normal, float* out,
                uint32_t out_w, uint32_t out_h)
{
    const size_t n = (size_t)in_w * in_h * 4;
    if (mode == 0) { if (out_w != in_w || out_h != in_h) return -1; memcpy(out, color, n * sizeof(float)); return 0; }
    if (mode == 1 && (out_w != in_w || out_h != in_h)) return -1;
    if (mode == 2 && (out_w / 2u != in_w || out_h / 2u != in_h)) return -1;
    if (mode != 1 && mode != 2) return -1;
    float* a = (float*)malloc(n * sizeof(float));
    float* b = (float*)malloc(n * sizeof(float));
    if (!a || !b) { free(a); free(b); return -2; }
    const float* src = color;
    float* pp[2] = { a, b };
    for (int it = 0; it < 5; it++) {
        float* dst = pp[it & 1];
        dn_atrous_pass(src, normal, albedo, dst, (int)in_w, (int)in_h, 1 << it, it < 2 ? 0.0f : 1.0f / (float)(1 << (it - 2)));
        src = dst;
    }
    if (mode == 1) memcpy(out, src, n * sizeof(float));
    else {
        const int iw = (int)in_w, ih = (int)in_h;
        for (int Y = 0; Y < (int)out_h; Y++)
            for (int X = 0; X < (int)out_w; X++) {
                const int x0 = (X & 1) ? (X >> 1) : (X >> 1) - 1, y0 = (Y & 1) ? (Y >> 1) : (Y >> 1) - 1;
                const float fx = (X & 1) ? 0.25f : 0.75f, fy = (Y & 1) ? 0.25f : 0.75f;
                int xa = x0 < 0 ? 0 : (x0 > iw - 1 ? iw - 1 : x0), xb = x0 + 1 < 0 ? 0 : (x0 + 1 > iw - 1 ? iw - 1 : x0 + 1);
                int ya = y0 < 0 ? 0 : (y0 > ih - 1 ? ih - 1 : y0), yb = y0 + 1 < 0 ? 0 : (y0 + 1 > ih - 1 ? ih - 1 : y0 + 1);
                const float gx = 1.0f - fx, gy = 1.0f - fy;
                for (int k = 0; k < 4; k++) {
                    const float pa = src[((size_t)ya * iw + xa) * 4 + k], pb = src[((size_t)ya * iw + xb) * 4 + k];
                    const float pc = src[((size_t)yb * iw + xa) * 4 + k], pd = src[((size_t)yb * iw + xb) * 4 + k];
                    out[((size_t)Y * out_w + X) * 4 + k] = (pa * gx + pb * fx) * gy + (pc * gx + pd * fx) * fy;
                }
            }
    }
    free(a); free(b);
    return 0;
}
