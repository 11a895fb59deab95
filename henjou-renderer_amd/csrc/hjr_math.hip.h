// Device-side fp32 toolkit for the Henjou hot path (gfx950).
//
// Numerics contract: this translation unit is compiled with -ffp-contract=off and HIP's default
// correctly-rounded fp32 divide/sqrt, so every + - * / sqrt below is one IEEE operation in the order written;
// fused multiply-adds appear only where fmaf() is spelled out.  Transcendental functions are built from those
// operations alone (Cephes single-precision polynomials), which makes every value bit-reproducible on any IEEE
// machine — the CPU oracle's PORTABLE mode re-derives the same bits independently.
//
// The vector helpers restate NVIDIA OptiX SDK 7.7 sutil/vec_math.h, which the reference's kernel headers are
// written against (float3 / float == multiply by 1.0f / s, normalize == v * (1.0f / sqrtf(dot(v, v))), ...).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define HD __device__ __forceinline__

struct f3 { float x, y, z; };
struct f2 { float x, y; };

HD f3 V(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
HD f3 V1(float s) { return V(s, s, s); }
HD f3 operator+(f3 a, f3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
HD f3 operator-(f3 a, f3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
HD f3 operator-(f3 a) { return V(-a.x, -a.y, -a.z); }
HD f3 operator*(f3 a, f3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
HD f3 operator*(f3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
HD f3 operator/(f3 a, float s) { float inv = 1.0f / s; return a * inv; }
HD f3 ssub(float s, f3 a) { return V(s - a.x, s - a.y, s - a.z); }
HD float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
HD f3 cross(f3 a, f3 b) { return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
HD float length3(f3 v) { return sqrtf(dot(v, v)); }
HD f3 normalize(f3 v) { float invLen = 1.0f / sqrtf(dot(v, v)); return v * invLen; }
HD f3 reflect3(f3 i, f3 n) { return i - (n * 2.0f) * dot(n, i); }
HD f3 lerp3(f3 a, f3 b, float t) { return a + (b - a) * t; }
HD float clampf(float f, float a, float b) { return fmaxf(a, fminf(f, b)); }
HD float absdot(f3 a, f3 b) { return fabsf(dot(a, b)); }

#define HJ_PI 3.14159265358979323846f
#define HJ_PI2 6.28318530717958647692f
#define HJ_INV_PI 0.31830988618379067154f
#define HJ_FLT_MAX 3.402823466e+38f
#define HJ_FLT_MIN 1.175494351e-38f

HD float bits2f(uint32_t u) { return __uint_as_float(u); }
HD uint32_t f2bits(float f) { return __float_as_uint(f); }

// Correctly rounded fp32 division spelled out with the hardware's division helpers (scale, reciprocal, Newton steps with fma, fixup): the
// sequence a compiler emits for `a / b` under -fhip-fp32-correctly-rounded-divide-sqrt.  The ray / triangle test uses it, so that its bits
// do not depend on the flags of the translation unit (the HJR_FAST_MATH units compile `/` as a multiplication by v_rcp_f32).
HD float exact_div(float a, float b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    bool vcc;
    const float ds = __builtin_amdgcn_div_scalef(a, b, false, &vcc); // the denominator, scaled
    const float ns = __builtin_amdgcn_div_scalef(a, b, true, &vcc);  // the numerator, scaled; vcc: the quotient needs the final scaling
    float r = __builtin_amdgcn_rcpf(ds);
    const float e = __builtin_fmaf(-ds, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    float q = ns * r;
    float t = __builtin_fmaf(-ds, q, ns);
    q = __builtin_fmaf(t, r, q);
    t = __builtin_fmaf(-ds, q, ns);
    return __builtin_amdgcn_div_fixupf(__builtin_amdgcn_div_fmasf(t, r, q, vcc), b, a);
#else
    return a / b;
#endif
}

// ---- portable transcendental set (exact kernels); HJR_FAST_MATH translation units take the hardware's approximations instead
#ifdef HJR_FAST_MATH
#define HJR_FAST_SINCOS(x, s, c) { s = __sinf(x); c = __cosf(x); }
#endif
HD void p_sincos(float x, float& s, float& c)
{
#ifdef HJR_FAST_MATH
    HJR_FAST_SINCOS(x, s, c)
    return;
#endif
    float fj = floorf(x * 0.636619772367581343f + 0.5f);
    int j = (int)fj;
    float r = fmaf(fj, -1.5703125f, x);
    r = fmaf(fj, -4.837512969970703125e-4f, r);
    r = fmaf(fj, -7.54978995489188216e-8f, r);
    float z = r * r;
    float sp = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f) * z, r, r);
    float cp = fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f), z * z,
                    fmaf(-0.5f, z, 1.0f));
    int q = j & 3;
    float ss = (q & 1) ? cp : sp;
    float cc = (q & 1) ? sp : cp;
    s = (q & 2) ? -ss : ss;
    c = (q == 1 || q == 2) ? -cc : cc;
}
HD float p_asin_poly(float x, float z)
{
    float p = fmaf(fmaf(fmaf(fmaf(4.2163199048e-2f, z, 2.4181311049e-2f), z, 4.5470025998e-2f), z, 7.4953002686e-2f), z,
                   1.6666752422e-1f);
    return fmaf(p * z, x, x);
}
HD float p_acos(float x)
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
HD float p_log(float x)
{
    uint32_t u = f2bits(x);
    int e = (int)(u >> 23) - 126;
    float m = bits2f((u & 0x007fffffu) | 0x3f000000u);
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
HD float p_exp(float x)
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
HD float p_pow(float x, float y)
{
#ifdef HJR_FAST_MATH
    return __powf(x, y); // v_log_f32, multiply, v_exp_f32
#endif
    if (y == 0.0f) return 1.0f;
    if (x == 1.0f) return 1.0f;
    if (x != x || y != y) return x + y;
    if (x < 0.0f) return bits2f(0x7fc00000u);
    if (x == 0.0f) return (y > 0.0f) ? 0.0f : bits2f(0x7f800000u);
    if (x > HJ_FLT_MAX) return (y > 0.0f) ? x : 0.0f;
    float lx = (x < HJ_FLT_MIN) ? p_log(x * 16777216.0f) - 16.6355323334f : p_log(x);
    float t = y * lx;
    if (t != t) return t;
    if (t > 88.7f) return bits2f(0x7f800000u);
    if (t < -87.0f) return 0.0f;
    return p_exp(t);
}
HD float p_pow5(float x) { float x2 = x * x; float x4 = x2 * x2; return x4 * x; }
// Cephes atanf on [0, inf) + quadrant logic (equirect sky lookup only)
HD float p_atan_pos(float x)
{
    float y0;
    if (x > 2.414213562373095f) { y0 = 1.57079632679489661923f; x = -(1.0f / x); }
    else if (x > 0.4142135623730950f) { y0 = 0.78539816339744830962f; x = (x - 1.0f) / (x + 1.0f); }
    else y0 = 0.0f;
    float z = x * x;
    float p = fmaf(fmaf(fmaf(8.05374449538e-2f, z, -1.38776856032e-1f), z, 1.99777106478e-1f), z, -3.33329491539e-1f);
    return y0 + fmaf(p * z, x, x);
}
HD float p_atan2(float y, float x)
{
    if (x != x || y != y) return x + y;
    if (y == 0.0f) return (x >= 0.0f && !(f2bits(x) >> 31)) ? y : copysignf(HJ_PI, y);
    if (x == 0.0f) return copysignf(1.57079632679489661923f, y);
    float a = p_atan_pos(fabsf(y / x));
    if (x < 0.0f) a = HJ_PI - a;
    return copysignf(a, y);
}
