// Material side of the hot path: surface record, thin-film LUT and texture fetches, DisneyBRDF (kernel/disneyBRDF.h),
// MetaMaterialGlass and the multiple-scattering GGX walk (kernel/BSDFs.h), and the BSDF dispatch.
#pragma once
#include "hjr_sampling.hip.h"

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
    float lambda_wo; // d_Lambda(alpha, wo): evaluate (G2), getPDFSpecular (G1) and the evaluate inside sample all need it for the same wo — computed once per shaded hit (disney_lambda_wo)
    bool is_thinfilm;
};
HD Disney disney_init(const Surface& s, float lambda_wo) // :165-177
{
    Disney d;
    d.lambda_wo = lambda_wo;
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
HD float d_G1(float lambda_w) { return 1.0f / (1.0f + lambda_w); }                                             // :50-52
HD float d_G2(float a, f3 wi, float lambda_wo) { return 1.0f / (1.0f + d_Lambda(a, wi) + lambda_wo); }            // :54-56
HD float disney_lambda_wo(const Surface& s, f3 wo) { return d_Lambda(clampf(s.roughness * s.roughness, 0.01f, 1.0f), wo); }
HD float d_getPDFDiffuse(f3 wi) { return fabsf(wi.y) * HJ_INV_PI; }                                  // :40-42
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
HD float d_getPDFSpecular(float a, f3 wm, f3 wo, float lambda_wo) // :88-90
{
    return 0.25f * ggx_D(a, wm) * d_G1(lambda_wo) * absdot(wo, wm) / (absdot(wm, wo) * fabsf(wo.y));
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
// The subsurface, sheen and clearcoat terms enter the result multiplied by 0 when the material's sheen and clearcoat are +0
// (every material of the glTF path unless an extension sets them; m_subsurface is always 0): each term is then +0 exactly as
// long as it is finite, which min(|wo.y|, |wi.y|) > 1e-15 guarantees (every denominator below is >= 1e-30, every numerator
// < 1e5, and the clearcoat factors are positive).  A wave whose lanes are ALL in that case adds the literal +0 instead of
// computing them (8 correctly rounded divisions and 2 square roots per evaluation); one lane outside it (a grazing direction:
// the recorded NaN samples of the C2 frame, or a material with sheen / clearcoat) sends the whole wave through the full
// expression.  Same bits either way.
HD bool disney_plain(const Disney& d, f3 wo, f3 wi)
{
    const bool plain = __float_as_uint(d.sheen) == 0u && __float_as_uint(d.clearcoat) == 0u && fminf(fabsf(wo.y), fabsf(wi.y)) > 1e-15f;
    return __ballot(!plain) == 0ull;
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
    const bool plain = disney_plain(d, wo, wi);
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
        float ggxG = d_G2(d.alpha, wi, d.lambda_wo);
        f3 ggxF = schlick3(F0, wo, wm);
        f_specular = (ggxF * 0.25f * ggxD * ggxG) / (fabsf(wo.y) * fabsf(wi.y));
    }
    if (plain) return ((f_diffuse + V1(0.0f)) + V1(0.0f)) * (1.0f - d.metallic) + f_specular + V1(0.0f);
    float deltacos = 1.0f / (dot_wi_n + dot_wo_n) - 0.5f;
    f3 f_subsurface = d.basecolor * HJ_INV_PI * 1.25f * (f_tsi * f_tso * deltacos + 0.5f);
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
// Lobe weights of sample() / getPDF() (:238-246, :310-316): three divisions by 1.5 - metallic.  A wave whose lanes all have metallic = 0 (every
// non-metal of a glTF scene; metals above 0.5 take the multiple-scattering GGX lobe) uses the quotients of the constants: 1.0f * (1.0f - 0.0f) = 1,
// (1 + 0.5) + 0 = 1.5, and 1 / 1.5, 0.5 / 1.5, 0 / 1.5 rounded as the correctly rounded division rounds them.
HD void disney_weights(const Disney& d, float& dw, float& sw, float& cw)
{
    if (__ballot(d.metallic != 0.0f) == 0ull) {
        constexpr float k_dw = 1.0f / 1.5f, k_sw = 0.5f / 1.5f, k_cw = 0.0f / 1.5f;
        dw = k_dw; sw = k_sw; cw = k_cw;
        return;
    }
    float diffuseWeight = 1.0f * (1.0f - d.metallic);
    float specularWeight = 0.5f;
    float clearcoatWeight = 0.0f;
    float sumWeight = diffuseWeight + specularWeight + clearcoatWeight;
    dw = diffuseWeight / sumWeight;
    sw = specularWeight / sumWeight;
    cw = clearcoatWeight / sumWeight;
}
HD f3 disney_sample(const KParams& P, const Disney& d, f3 wo, f3& wi, float& pdf, CMJState& st) // :237-307
{
    float dw, sw, cw;
    disney_weights(d, dw, sw, cw);
    float select_p = cmj_1d(st);
    float pdf_diffuse = 1.0f, pdf_specular = 1.0f, pdf_clearcoat = 1.0f;
    f2 xi = cmj_2d(st);
    // The reference's three branches (:262-291) repeat most of their work: the azimuth sin / cos, the normalisation of the half
    // vector (diffuse and specular lobe) and the three lobe pdfs are the same operations on the lane's own (xi, wi, wm) whichever lobe
    // was chosen.  A wave that holds lanes of both the diffuse and the specular lobe (two thirds / one third for a dielectric) runs
    // those parts once here instead of once per branch; each lane still executes exactly the reference's operation sequence on its
    // own values.  (The clearcoat lobe has weight 0: it is only reached when dw + sw rounds below 1 and select_p falls into the gap.)
    const int lobe = select_p < dw ? 0 : (select_p < dw + sw ? 1 : 2);
    const float phi = 2.0f * HJ_PI * (lobe == 1 ? xi.x : xi.y); // d_sampleDiffuse :32, d_sampleClearcoat :97 (HJ_PI2 == 2 HJ_PI) / sample_visible_normal :68
    float sp, cp;
    p_sincos(phi, sp, cp);
    f3 v = V(0.0f, 1.0f, 0.0f), wm = v; // v: the half vector before its normalisation (diffuse, specular)
    if (lobe == 0) { // d_sampleDiffuse (:30-38), then wm = normalize(wi + wo)
        const float theta = 0.5f * p_acos(1.0f - 2.0f * xi.x);
        float sinTheta, cosTheta;
        p_sincos(theta, sinTheta, cosTheta);
        wi = V(cp * sinTheta, cosTheta, sp * sinTheta);
        v = wi + wo;
    } else if (lobe == 1) { // sample_visible_normal (:64-80)
        const f3 strech_wo = normalize(V(wo.x * d.alpha, wo.y, wo.z * d.alpha));
        const float z = fmaf((1.0f - xi.y), (1.0f + strech_wo.y), -strech_wo.y);
        const float sinTheta = sqrtf(clampf(1.0f - z * z, 0.0f, 1.0f));
        const f3 h = V(cp * sinTheta, z, sp * sinTheta) + strech_wo;
        v = V(h.x * d.alpha, h.y, h.z * d.alpha);
    } else { // d_sampleClearcoat (:93-100): wm as sampled, not normalised again
        const float ca = HJ_CLEARCOAT_ALPHA;
        const float cosineTheta = sqrtf(fmaxf((1.0f - p_pow(ca * ca, 1.0f - xi.x)) / (1.0f - ca * ca), 0.0f));
        const float sinTheta = sqrtf(fmaxf(1.0f - cosineTheta * cosineTheta, 0.0f));
        wm = V(cp * sinTheta, cosineTheta, sp * sinTheta);
    }
    if (lobe != 2) wm = normalize(v);
    if (lobe != 0) wi = reflect3(-wo, wm);
    pdf_diffuse = d_getPDFDiffuse(wi);
    pdf_specular = d_getPDFSpecular(d.alpha, wm, wo, d.lambda_wo);
    // cw is +0: cw * pdf_clearcoat is +0 whenever pdf_clearcoat is finite, i.e. |wm . wo| is not vanishingly small (its numerator is
    // < 1e5 and not negative); a wave whose lanes all satisfy that adds the literal (same bits), else the wave computes the term.
    if (__ballot(!(absdot(wm, wo) > 1e-30f)) == 0ull) pdf = (dw * pdf_diffuse + sw * pdf_specular) + 0.0f;
    else {
        pdf_clearcoat = d_getPDFClearcoat(wm, wo);
        pdf = dw * pdf_diffuse + sw * pdf_specular + cw * pdf_clearcoat;
    }
    if (wi.y < 0.0f) { pdf = 1.0f; return V1(0.0f); }
    return disney_eval(P, d, wo, wi);
}
HD float disney_pdf(const Disney& d, f3 wo, f3 wi) // :309-326
{
    float dw, sw, cw;
    disney_weights(d, dw, sw, cw);
    f3 wm = normalize(wo + wi);
    return dw * d_getPDFDiffuse(wi) + sw * d_getPDFSpecular(d.alpha, wm, wo, d.lambda_wo);
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
// lambda_wo = disney_lambda_wo(s, wo), computed once by the caller for all the calls of one shaded hit
HD f3 bsdf_eval(const KParams& P, const Surface& s, f3 wo, f3 wi, float lambda_wo)
{
    if (s.is_specular) return V1(0.0f);
    Disney d = disney_init(s, lambda_wo);
    return disney_eval(P, d, wo, wi);
}
HD f3 bsdf_sample(const KParams& P, const Surface& s, f3 wo, f3& wi, float& pdf, CMJState& st, float lambda_wo)
{
    if (s.is_specular) return metaglass_sample(s.ior, wo, wi, pdf, st);
    if (!(s.metallic > 0.5f)) {
        Disney d = disney_init(s, lambda_wo);
        return disney_sample(P, d, wo, wi, pdf, st);
    }
    return msggx_sampleBSDF(s.basecolor, clampf(s.roughness * s.roughness, 0.0001f, 1.0f), wo, wi, st, pdf);
}
HD float bsdf_pdf(const Surface& s, f3 wo, f3 wi, float lambda_wo)
{
    if (s.is_specular) return 0.0f;
    Disney d = disney_init(s, lambda_wo);
    return disney_pdf(d, wo, wi);
}
