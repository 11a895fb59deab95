/*
 * hjr_oracle.h — CPU ORACLE for the Henjou per-pixel-sample hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product (libhenjou_hip.so) never links,
 * includes or calls anything in oracle/.
 *
 * What it is: a plain-C restatement of the reference's device algorithm
 *   (/root/reference/include/kernel/{cmj,math,BSDFs,disneyBRDF,light_sample,rt}.h)
 * plus the build-defined entry points that are missing from the reference
 * (raygen / closest-hit / miss — SURVEY.md §0 F1, §8a rows a1,a4,a5).
 *
 * Pinning status: PARTIALLY PINNED.  The reference has no tests and cannot be built here
 * (needs the OptiX SDK, sutil, and its own missing henjouRenderer.h).  The only reference-derived
 * numbers available are the known-answer values recorded in SURVEY.md §8c (CMJ, math helpers, one
 * Disney / glass / msGGX call each); tests/test_oracle_kat.py checks this file against every one of
 * them.  Everything without such a value (NEE radiance, ray traversal, raygen, LUT contents) is
 * "parity unpinned" — see DESIGN.md §3.
 *
 * Three math back-ends (selected per context / per call):
 *   HJO_MATH_LIBM     : glibc sinf/cosf/acosf/powf — the arithmetic the SURVEY §8c values came from.
 *   HJO_MATH_PORTABLE : transcendental functions built only from IEEE + - * / sqrt fma, identical
 *                       operation-for-operation to the HIP kernel's, so GPU == oracle bit-for-bit.
 *   HJO_MATH_HOSTF64  : LIBM, but un-suffixed sin/cos/acos/pow/sqrt/fma calls evaluated in double, as in the
 *                       g++ host compile that produced SURVEY §8c's values (KAT pinning only).
 */
#ifndef HJR_ORACLE_H
#define HJR_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { HJO_MATH_LIBM = 0, HJO_MATH_PORTABLE = 1, HJO_MATH_HOSTF64 = 2 };
enum { HJO_INTEGRATOR_NEE = 0, HJO_INTEGRATOR_PT = 1, HJO_INTEGRATOR_MIS = 2 };

/* Mirror of the reference's HitGroupData (renderer.h:659-687), texture slots dropped. */
typedef struct hjo_material {
    float basecolor[3];
    float metallic;
    float roughness;
    float sheen;
    float clearcoat;
    float ior;
    float transmission;
    float emission[3];
    int32_t is_light;
    int32_t ideal_specular;
    int32_t is_thinfilm;
    int32_t basecolor_tex;          /* texture slot or -1 */
    int32_t metallic_roughness_tex; /* slot or -1: G = roughness, B = metallic */
    int32_t normal_tex, emission_tex, _reserved;
} hjo_material; /* 80 bytes */

typedef struct hjo_texture { const uint8_t* rgba8; uint32_t width, height; int32_t srgb; int32_t _reserved; } hjo_texture;

/* Mirror of SceneData (scene.h:19-36) + the per-frame Matrix4x3 arrays (renderer.h:257-291). */
typedef struct hjo_scene {
    uint32_t n_tris, n_instances, n_materials, n_lights;
    const float*    vertices;      /* 9*n_tris, object space, de-indexed */
    const float*    normals;       /* 9*n_tris */
    const float*    texcoords;     /* 6*n_tris */
    const uint32_t* indices;       /* 3*n_tris (== 0,1,2,... on the glTF path) */
    const uint32_t* material_ids;  /* n_tris */
    const uint32_t* prim_offsets;  /* n_instances, first global triangle of each instance */
    const float*    transforms;    /* 12*n_instances, row-major 3x4 */
    const float*    inv_transforms;/* 12*n_instances */
    const hjo_material* materials;
    const uint32_t* light_prim_ids;      /* n_lights, global triangle ids */
    const float*    light_prim_emission; /* 3*n_lights */
    const uint8_t*  lut_rgba;      /* thin-film LUT, may be NULL */
    int32_t lut_w, lut_h;
    const hjo_texture* textures;   /* material texture slots */
    uint32_t n_textures;
    int32_t sky_w, sky_h;          /* equirect IBL (float RGBA), 0 = constant sky from hjo_params.sky */
    const float*    sky_rgba;
} hjo_scene;

typedef struct hjo_params {
    uint32_t width, height, spp, frame, seed, integrator;
    float cam_pos[3], cam_dir[3], cam_up[3], cam_right[3];
    float cam_f;
    float sky[3];            /* scene_sky_default */
    float ibl_intensity;
    uint32_t x0, y0, x1, y1; /* half-open pixel rectangle to render; x1==0 => full frame */
} hjo_params;

typedef struct hjo_stats {
    uint64_t samples, closest_rays, shadow_rays, box_tests_closest, tri_tests_closest,
             box_tests_shadow, tri_tests_shadow, shaded_hits, light_samples, nan_samples;
} hjo_stats;

typedef struct hjo_ctx hjo_ctx;

/* Builds world-space triangles + a BVH for the given scene/transforms. */
hjo_ctx* hjo_create(const hjo_scene* scene, int math_mode);
void     hjo_destroy(hjo_ctx*);

/* Full render: color/albedo/normal are width*height*4 floats (row y, column x, pix = x + y*width). */
int hjo_render(hjo_ctx*, const hjo_params*, float* color, float* albedo, float* normal,
               int nthreads, hjo_stats* stats);
/* One (pixel, sample): radiance[3], albedo[3], normal[3]. */
/* narrate NEE samples on stderr (test aid) */
void hjo_set_trace(int on);
int hjo_sample(hjo_ctx*, const hjo_params*, uint32_t x, uint32_t y, uint32_t s,
                float* radiance, float* albedo, float* normal);

/* Ray queries (bvh=0: brute force over all triangles; bvh=1: through the BVH).
 * closest: returns global prim id or -1; out = {t, b1, b2}. */
int hjo_trace_closest(hjo_ctx*, const float* o, const float* d, float tmin, float tmax, int use_bvh, float* out);
int hjo_trace_any(hjo_ctx*, const float* o, const float* d, float tmin, float tmax, int use_bvh);

/* ---- known-answer-test entry points (one per reference function) ---- */
uint32_t hjo_xxhash32_u4(uint32_t x, uint32_t y, uint32_t z, uint32_t w);      /* cmj.h:38-51  */
uint32_t hjo_cmj_permute(uint32_t i, uint32_t l, uint32_t p);                   /* cmj.h:60-91  */
float    hjo_cmj_randfloat(uint32_t i, uint32_t p);                             /* cmj.h:93-106 */
void     hjo_cmj(uint32_t index, uint32_t scramble, float* out2);               /* cmj.h:108-117*/
/* state = {n_spp_lo, n_spp_hi, scramble, depth, image_idx}; advanced in place */
void     hjo_cmj_2d(uint32_t* state5, float* out2);                             /* cmj.h:119-128*/
void     hjo_cosine_sampling(int math_mode, float u, float v, float* wi3, float* pdf); /* math.h:7-15 */
void     hjo_orthonormal_basis(const float* n3, float* t3, float* b3);          /* math.h:43-51 */
int      hjo_refract(const float* v3, const float* n3, float ior1, float ior2, float* r3); /* math.h:92-103 */
float    hjo_schlick_ior(int math_mode, float no, float ni, const float* w3, const float* n3); /* math.h:31-37 */
/* BSDF layer.  mat: material; wo local (y-up). state5 as above.  mode 0=Disney 1=glass 2=msGGX 3=dispatch */
void     hjo_bsdf_sample(int math_mode, const hjo_material* mat, int which, const float* wo3,
                         uint32_t* state5, float* f3out, float* wi3, float* pdf);
void     hjo_bsdf_eval(int math_mode, const hjo_material* mat, const float* wo3, const float* wi3,
                       const uint8_t* lut, int lut_w, int lut_h, float* f3out);
float    hjo_bsdf_pdf(int math_mode, const hjo_material* mat, const float* wo3, const float* wi3);
/* portable math, for ulp comparisons against libm */
float hjo_p_sin(float x); float hjo_p_cos(float x); float hjo_p_acos(float x);
float hjo_p_pow(float x, float y); float hjo_p_pow5(float x); float hjo_p_atan2(float y, float x);
/* 8-bit RGBA texture fetch: wrap, bilinear (CUDA weights), optional sRGB decode before filtering */
void hjo_tex_fetch(const uint8_t* rgba, int w, int h, int srgb, float u, float v, float* out3);
/* equirect sky lookup for a unit direction */
void hjo_sky_fetch(int math_mode, const float* rgba, int w, int h, const float* dir3, float* out3);
/* thin-film LUT lookup (disneyBRDF.h:11-14 + renderer.h:854-898 sampler state) */
void hjo_lut_fetch(const uint8_t* rgba, int w, int h, float u, float v, float* out3);
/* output stage (renderer.h:73-101) */
void hjo_float4_to_srgb8(const float* rgba, uint8_t* out, uint32_t n_pixels);
/* kernel/color.h tonemappers (1 = Uchimura :10-53, 2 = ACES :55-63) followed by the stage above */
void hjo_tonemap_to_srgb8(const float* rgba, uint8_t* out, uint32_t n_pixels, int mode);
/* denoise-mode replacement (see hjr_oracle.c): mode 0 Default (copy), 1 Denoise, 2 DenoiseUpScale2X; float4 images; 0 = ok */
int hjo_denoise(int mode, uint32_t in_w, uint32_t in_h, const float* color, const float* albedo, const float* normal, float* out,
                uint32_t out_w, uint32_t out_h);
float hjo_tonemap(float x, int mode);

#ifdef __cplusplus
}
#endif
#endif
