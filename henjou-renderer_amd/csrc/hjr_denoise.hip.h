// Denoise-mode replacement.  The reference runs the OptiX AI denoiser over (aov_color | guide albedo |
// guide normal) -> AOV_Output (renderer/denoiser.h:42-189, renderer/renderer.h:1093-1120, 1258-1270); that network is closed, so
// its pixels cannot be reproduced.  What is reproduced is the data flow of the three render modes (render_option.h:38-43):
//   Default            output = input                                   (blendFactor 1, denoiser.h:94-97)
//   Denoise            output = filter(color; albedo, normal), same size
//   DenoiseUpScale2X   rendered at (W/2, H/2) (renderer.h:1096-1099), filtered, then brought to W x H
// with an edge-avoiding a-trous wavelet filter (Dammertz et al., HPG 2010) guided by the same two AOVs, and a 2x bilinear
// upscale.  Build-defined, specified to the bit so that the test suite's CPU checker can restate it independently:
//   5 passes, tap distance 1, 2, 4, 8, 16; 5 x 5 taps with B3 weights h = (1/16, 1/4, 3/8, 1/4, 1/16), rows outer, clamped
//   to the frame; per tap  w = ((wc * wn) * wa) * (h[dy] * h[dx])  with  w? = min(p_exp(max(-q / phi, -87)), 1):
//     wn: q = |normal_c - normal_t|^2, phi 0.25;   wa: q = |albedo_c - albedo_t|^2, phi 0.05   (the two guide AOVs)
//     wc: passes 0 and 1: 1 (guides only: isolated fireflies are averaged away before the colour term can protect them);
//         passes 2, 3, 4: q = |colour_c - colour_t|^2 / (0.01 + s * s), s = (c.x + c.y) + c.z of the CURRENT centre colour,
//         phi = 1, 0.5, 0.25 (relative, so the tolerance follows the local radiance level);
//   |d|^2 = dx*dx + dy*dy + dz*dz; out.rgb = (sum of tap.rgb * w) / (sum of w), out.a = centre.a; fp32, no contraction,
//   p_exp = the portable exponential.  On the bundled scene against a 512 spp frame: RMSE 1.76 -> 0.15 at 4 spp, 0.32 -> 0.15
//   at 64 spp (light sources and background excluded).
// Streaming kernels, one lane per pixel: 75 float4 reads per pixel and pass (mostly L2 hits), bound by HBM/L2 bandwidth;
// < 1 % of a frame's render time.
#pragma once
#include "hjr_math.hip.h"

#define HJR_ATROUS_PASSES 5

__global__ void __launch_bounds__(256) hjr_atrous_kernel(const float4* __restrict__ in, const float4* __restrict__ normal,
                                                          const float4* __restrict__ albedo, float4* __restrict__ out,
                                                          int W, int H, int step, float c_phi)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const float hk[5] = { 0.0625f, 0.25f, 0.375f, 0.25f, 0.0625f };
    const float n_phi = 0.25f, a_phi = 0.05f;
    const size_t c = (size_t)y * W + x;
    const float4 c0 = in[c], n0 = normal[c], a0 = albedo[c];
    const float s0 = (c0.x + c0.y) + c0.z;
    const float rel = 0.01f + s0 * s0;
    float sx = 0.0f, sy = 0.0f, sz = 0.0f, cum = 0.0f;
    for (int j = -2; j <= 2; j++) {
        const int yy = min(max(y + j * step, 0), H - 1);
        for (int i = -2; i <= 2; i++) {
            const int xx = min(max(x + i * step, 0), W - 1);
            const size_t t = (size_t)yy * W + xx;
            const float4 ct = in[t], nt = normal[t], at = albedo[t];
            float dx = c0.x - ct.x, dy = c0.y - ct.y, dz = c0.z - ct.z;
            float d2 = dx * dx + dy * dy + dz * dz;
            float wc = 1.0f;
            if (c_phi > 0.0f) wc = fminf(p_exp(fmaxf(-(d2 / rel) / c_phi, -87.0f)), 1.0f);
            dx = n0.x - nt.x; dy = n0.y - nt.y; dz = n0.z - nt.z;
            d2 = dx * dx + dy * dy + dz * dz;
            const float wn = fminf(p_exp(fmaxf(-d2 / n_phi, -87.0f)), 1.0f);
            dx = a0.x - at.x; dy = a0.y - at.y; dz = a0.z - at.z;
            d2 = dx * dx + dy * dy + dz * dz;
            const float wa = fminf(p_exp(fmaxf(-d2 / a_phi, -87.0f)), 1.0f);
            const float w = ((wc * wn) * wa) * (hk[j + 2] * hk[i + 2]);
            sx = sx + ct.x * w; sy = sy + ct.y * w; sz = sz + ct.z * w;
            cum = cum + w;
        }
    }
    out[c] = make_float4(sx / cum, sy / cum, sz / cum, c0.w);
}

// 2x bilinear upscale, pixel centres: source coordinate (X + 0.5) / 2 - 0.5, i.e. weights 0.75 / 0.25 towards the nearer texel,
// indices clamped to the source frame; out = (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy per channel.
__global__ void __launch_bounds__(256) hjr_upscale2x_kernel(const float4* __restrict__ in, float4* __restrict__ out, int iw, int ih, int ow, int oh)
{
    const int X = blockIdx.x * 64 + (threadIdx.x & 63), Y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (X >= ow || Y >= oh) return;
    const int x0 = (X & 1) ? (X >> 1) : (X >> 1) - 1, y0 = (Y & 1) ? (Y >> 1) : (Y >> 1) - 1;
    const float fx = (X & 1) ? 0.25f : 0.75f, fy = (Y & 1) ? 0.25f : 0.75f;
    const int xa = min(max(x0, 0), iw - 1), xb = min(max(x0 + 1, 0), iw - 1);
    const int ya = min(max(y0, 0), ih - 1), yb = min(max(y0 + 1, 0), ih - 1);
    const float4 a = in[(size_t)ya * iw + xa], b = in[(size_t)ya * iw + xb], c = in[(size_t)yb * iw + xa], d = in[(size_t)yb * iw + xb];
    const float gx = 1.0f - fx, gy = 1.0f - fy;
    float4 r;
    r.x = (a.x * gx + b.x * fx) * gy + (c.x * gx + d.x * fx) * fy;
    r.y = (a.y * gx + b.y * fx) * gy + (c.y * gx + d.y * fx) * fy;
    r.z = (a.z * gx + b.z * fx) * gy + (c.z * gx + d.z * fx) * fy;
    r.w = (a.w * gx + b.w * fx) * gy + (c.w * gx + d.w * fx) * fy;
    out[(size_t)Y * ow + X] = r;
}
