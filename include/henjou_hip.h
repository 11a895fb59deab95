/*
 * henjou_hip.h — C-ABI of libhenjou_hip.so, the MI355X-native replacement for Henjou-Renderer's
 * per-pixel-sample hot path (ray generation -> BVH traversal -> BSDF -> NEE integrator).
 *
 * Plain C, no exceptions across the boundary, no torch / STL types in any signature.
 * Every entry point cites the reference interface it replaces (paths relative to the reference's
 * include/ directory).  Status codes: 0 = ok, negative = error; hjr_last_error() gives the text.
 *
 * The reference's in-process boundary is
 *     cudaMemcpy(d_param, &params) ; optixLaunch(pipeline, stream, d_param, sizeof(Params), &sbt, W, H, 1)
 * (renderer/renderer.h:1229-1242) against six OptiX programs that read `Params` (kernel/Payload.h:8-10)
 * and per-material HitGroupData records (renderer/renderer.h:647-738).  Its file-level surface is
 * render_option.json + a glTF scene in, <image_name>_<frame>.png out (renderer/renderer.h:1053-1317).
 * Both levels are exported here.
 */
#ifndef HENJOU_HIP_H
#define HENJOU_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define HJR_OK 0
#define HJR_ERR_ARG (-1)     /* bad argument / null pointer / size mismatch */
#define HJR_ERR_IO (-2)      /* file missing or unreadable */
#define HJR_ERR_PARSE (-3)   /* malformed JSON / glTF / PNG, or a mandatory key is absent */
#define HJR_ERR_DEVICE (-4)  /* HIP runtime error, or no gfx950 device / kernel image */
#define HJR_ERR_STATE (-5)   /* call order violated (render before upload, ...) */

/* ---- Sized structs (the rule that keeps callers and library binary-compatible across releases) --------------------------------
 * hjr_scene_view, hjr_render_option, hjr_params and hjr_stats may GROW at their end in later releases.  Each starts with
 * `struct_size`: the CALLER sets it to sizeof(its own struct) before handing the struct to ANY entry point, input or output
 * (HJR_INIT does it together with the zero fill).  The library copies min(struct_size, its own sizeof) bytes in either direction:
 *   - a field the caller's (older, shorter) struct does not have is never written and reads as 0, which selects the default;
 *   - a field a newer caller has and this library does not know is left untouched on output and ignored on input;
 *   - struct_size == 0 (a struct that was zero-filled but not initialised) is rejected with HJR_ERR_ARG.
 * The reference passes its launch block the same way: optixLaunch(..., d_param, sizeof(Params), ...) (renderer/renderer.h:1241).
 * hjr_material, hjr_texture and hjr_camera are array elements / embedded records of fixed layout (reserved words inside). */
#define HJR_INIT(s) do { memset(&(s), 0, sizeof(s)); (s).struct_size = (uint32_t)sizeof(s); } while (0) /* needs <string.h> */

enum { HJR_INTEGRATOR_NEE = 0, HJR_INTEGRATOR_PT = 1, HJR_INTEGRATOR_MIS = 2 }; /* kernel/rt.h:162,85,284 */
enum { HJR_MODE_DEFAULT = 0, HJR_MODE_DENOISE = 1, HJR_MODE_DENOISE_UPSCALE2X = 2, HJR_MODE_DEBUG = 3 }; /* renderer/render_option.h:38-43 */

/* Material record of the hot path: the fields of HitGroupData (renderer/renderer.h:659-687, filled from Material,
 * renderer/material.h:10-63) that the kernel-side code under include/kernel/ reads, in a 16-byte-aligned 80-byte layout.
 * It is NOT field-for-field HitGroupData; a binding fills it per material like this (INTEGRATION.md has the code):
 *   kept as is : basecolor, metallic, roughness, sheen, clearcoat, ior, transmission, emmision -> emission, is_light,
 *                ideal_specular, is_thinfilm, basecolor_tex, normal_tex, emmision_tex -> emission_tex
 *   merged     : metallic_tex + roughness_tex -> metallic_roughness_tex (the glTF loader always binds the same image to both,
 *                gltfloader.h:1144-1156; G = roughness, B = metallic)
 *   dropped    : specular (float3; no kernel header reads it), clearcoat_tex, bump_tex (bound by the host, read only by the
 *                missing closest-hit program; the clearcoat factor itself is kept)
 * normal_tex is sampled (tangent-space normal map, build-defined frame: DESIGN.md §10); emission_tex is always -1 on the glTF path. */
typedef struct hjr_material {
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
    int32_t basecolor_tex;   /* texture slot or -1 (Material.base_color_tex, TexType::sRGB, gltfloader.h:1133-1140) */
    int32_t metallic_roughness_tex; /* slot or -1 (metallic_tex == roughness_tex, NonColor, gltfloader.h:1144-1156): G = roughness, B = metallic */
    int32_t normal_tex;      /* slot or -1 (Material.normal_tex, NonColor, gltfloader.h:1168-1175): tangent-space normal map, per-triangle tangent frame (build-defined) */
    int32_t emission_tex;    /* always -1 on the glTF path (gltfloader.h:1159) */
    int32_t _reserved;
} hjr_material;              /* 80 bytes */

/* Texture(filename, type) — renderer/texture.h:16-39: 8-bit RGBA, bound wrap + linear + normalised coords, sRGB decode
 * unless NonColor (renderer.h:740-800). */
typedef struct hjr_texture {
    const uint8_t* rgba8;    /* width * height * 4, row 0 = top image row */
    uint32_t width, height;
    int32_t srgb;            /* 1: TexType::sRGB, 0: NonColor */
    int32_t _reserved;
} hjr_texture;

/* Borrowed, read-only view of SceneData (renderer/scene.h:19-36).  The library copies on upload. */
typedef struct hjr_scene_view {
    uint32_t struct_size;    /* sizeof(hjr_scene_view) of the caller (HJR_INIT) */
    uint32_t n_vertices;     /* == 3 * n_triangles on the glTF path (de-indexed, gltfloader.h:1484-1492) */
    uint32_t n_triangles;
    uint32_t n_instances;    /* instance i <-> geometry i, 1:1 (gltfloader.h:1507-1512) */
    uint32_t n_materials;
    uint32_t n_lights;       /* emissive triangles (gltfloader.h:1496-1500) */
    uint32_t n_animations;   /* == number of glTF nodes */
    uint32_t n_textures;     /* SceneData.textures (de-duplicated by file name, texture_load.h:7-20) */
    const float*    vertices;            /* float3 x n_vertices, object space */
    const float*    normals;             /* float3 x n_vertices */
    const float*    texcoords;           /* float2 x n_vertices */
    const uint32_t* indices;             /* 3 x n_triangles */
    const uint32_t* material_ids;        /* n_triangles */
    const uint32_t* prim_offset;         /* n_instances: first global triangle of the instance */
    const uint32_t* geometry_index_offset; /* n_instances (GeometryData.index_offset, scene.h:9-12) */
    const uint32_t* geometry_index_count;  /* n_instances */
    const uint32_t* instance_animation_id; /* n_instances (InstanceData.animation_id, scene.h:14-17) */
    const hjr_material* materials;
    const uint32_t* light_prim_ids;      /* n_lights, global triangle ids */
    const float*    light_prim_emission; /* float3 x n_lights */
    const hjr_texture* textures;         /* n_textures */
} hjr_scene_view;

/* Mirror of RenderOption (renderer/render_option.h:45-84). */
typedef struct hjr_render_option {
    uint32_t struct_size;        /* sizeof(hjr_render_option) of the caller (HJR_INIT) */
    uint32_t image_width, image_height;
    char image_name[256];
    char image_directory[512];
    uint32_t max_spp;
    char gltf_path[512];
    char gltf_name[256];
    uint32_t fps, start_frame, end_frame;
    float time_limit;
    int32_t allow_camera_animation;
    float camera_fov;            /* radians after load (render_json_loader.h:144) */
    float camera_position[3];
    float camera_direction[3];
    int32_t camera_animation_id; /* -1 = none */
    int32_t render_mode;         /* HJR_MODE_* */
    char ptxfile_path[512];      /* parsed, unused */
    int32_t use_IBL;
    char IBL_path[512];
    float IBL_intensity;
    float scene_sky_default[3];
    int32_t use_date, save_renderOption;
    char LUT_path[512];
    /* optional "Henjou_HIP" section (ignored by the reference): */
    uint32_t seed;               /* default 1 */
    int32_t integrator;          /* default HJR_INTEGRATOR_NEE */
    uint32_t devices;            /* default 1: GPUs of the node that share each frame (pixel-tile shard, one process per GPU) */
    uint32_t tile;               /* shard granularity in pixels; 8 is the only supported value */
    int32_t serial_io;           /* default 0; 1: hjr_render_file renders, writes and prepares the next frame one after the other (no overlap) */
    int32_t fast_math;           /* default 0; 1: hjr_render_file / henjou_cli launch with HJR_FLAG_FAST_MATH */
    int32_t force_rebuild;       /* default 0; 1: hjr_render_file rebuilds the frame data every frame even when nothing moved (benchmarking) */
} hjr_render_option;

typedef struct hjr_camera {      /* Params.camera_* (renderer/renderer.h:1187-1191) */
    float pos[3], dir[3], up[3], right[3];
    float f;                     /* camera_f = 2 / tan(fov) (renderer/renderer.h:1147) */
} hjr_camera;

/* Per-launch parameters: the scalar part of `Params` (renderer/renderer.h:1175-1227). */
typedef struct hjr_params {
    uint32_t struct_size;        /* sizeof(hjr_params) of the caller (HJR_INIT) */
    uint32_t width, height;      /* params.image_width/height */
    uint32_t spp;                /* params.spp */
    uint32_t frame;              /* params.frame */
    uint32_t seed;               /* CMJState.scramble (build-defined) */
    uint32_t integrator;         /* HJR_INTEGRATOR_* */
    hjr_camera camera;
    float sky[3];                /* scene_sky_default: the 1x1 IBL texel (renderer/texture.h:58-65) */
    float ibl_intensity;         /* params.ibl_intensity */
    uint32_t rank, world_size;   /* pixel-tile shard: this launch renders the 8x8 tiles whose id t (see hjr_owned_tiles) has t % world_size == rank */
    uint32_t flags;              /* HJR_FLAG_* */
} hjr_params;
#define HJR_FLAG_STATS 1u        /* run the counting variant of the kernel (slower; fills hjr_stats) */
#define HJR_FLAG_ZERO_UNOWNED 2u /* clear pixels of tiles this rank does not own (for a sum-reduce exchange) */
#define HJR_FLAG_PACKED 4u       /* the AOV buffers (device or host) are PACKED — this rank's tiles only, back to back, each
                                  * tile 64 float4 in row-major 8x8 order (hjr_owned_tiles(..) x 64 float4 per AOV).  What a
                                  * multi-GPU frame exchanges: 1 / world_size of the frame per rank, no zero fill */

#define HJR_FLAG_FAST_MATH 8u    /* opt-in: the approximate-arithmetic kernels — hardware reciprocal / square root / sine / cosine / power and fused
                                  * multiply-adds in the SHADING code, as the reference's own build does (nvcc --use_fast_math: div.approx, sqrt.approx,
                                  * sin.approx in lib/ptx); traversal and ray / triangle test unchanged.  Frames are NOT bit-identical to the default
                                  * (exact) kernels: they agree within the metric's tolerance (per-pixel RMSE < 1e-3 at 1024 spp; tests/test_gpu_fast_math.py).
                                  * Megakernel family only (NEE and Pathtrace); ignored by HJR_FLAG_STATS launches and by MIS launches, whose exact wavefront
                                  * kernels are faster than an approximate megakernel would be (hjr_stats.fast_math tells what ran). */

typedef struct hjr_stats {
    uint32_t struct_size;        /* sizeof(hjr_stats) of the caller (HJR_INIT) */
    uint32_t _pad0;
    uint64_t samples, closest_rays, shadow_rays, box_tests_closest, tri_tests_closest,
             box_tests_shadow, tri_tests_shadow, shaded_hits, light_samples, nan_samples;
    float    last_kernel_ms;     /* HIP-event time of the last render kernel on its stream */
    uint32_t bvh_nodes, bvh_depth, n_triangles;
    /* which megakernel layout the current frame data selects (csrc/hjr_launch.hip.h::hjr_launch): 0 = BVH4 read from memory,
     * 1 = BVH2 staged in LDS with 32-bit stack entries, 2 = BVH2 in LDS with 16-bit stack entries, 3 = BVH2 read from memory */
    uint32_t lds_mode;
    uint32_t stack_need;         /* worst-case traversal stack entries per lane of the current BVH */
    uint32_t stack_lds_entries;  /* memory-path layouts: entries of a lane's stack kept in LDS; deeper ones overflow to HBM */
    uint32_t pipeline;           /* 0 = persistent megakernel (a lane owns a path), 1 = workgroup-local wavefront kernel (trace / shade batches) */
    uint64_t stack_overflow_pushes; /* HJR_FLAG_STATS launches: stack pushes that went to the HBM overflow (memory-path layouts) */
    /* HJR_FLAG_STATS launches: where NaN / Inf samples came from (they are zeroed and counted in nan_samples; the reference has no
     * guard and would emit a NaN pixel): the first nan_located <= 8 of them in no particular order, as (pixel x, pixel y, sample) */
    uint32_t nan_located;
    uint32_t fast_math;          /* 1: the last launch ran the HJR_FLAG_FAST_MATH kernels */
    uint32_t nan_where[8][3];
} hjr_stats;

typedef struct hjr_scene hjr_scene; /* owning, host side (SceneData + animations) */
typedef struct hjr_ctx hjr_ctx;     /* one per device */

const char* hjr_last_error(void);

/* ---------------- scene surface: the file-level drop-in (host only, no GPU needed) ---------------- */
/* load_json(filepath, RenderOption&) — loader/render_json_loader.h:78-228 (incl. ./fps.txt override, :164-171).
 * `out` must carry its struct_size (HJR_INIT) before the call, like every sized struct. */
int hjr_load_render_option(const char* json_path, hjr_render_option* out);
/* gltfloader(filepath, filename, SceneData&, RenderOption&) — loader/gltfloader.h:1068-1601 */
int hjr_scene_load_gltf(const char* dir, const char* file, hjr_render_option* opt_inout, hjr_scene** out);
void hjr_scene_free(hjr_scene*);
int hjr_scene_get_view(const hjr_scene*, hjr_scene_view* out);
/* Renderer::updateIASMatrix(time) — renderer/renderer.h:257-291: per instance Matrix4x3 + inverse, row-major 3x4 */
int hjr_scene_eval_transforms(const hjr_scene*, float time, float* transforms12, float* inv_transforms12);
/* camera block of the frame loop — renderer/renderer.h:1145-1169 */
int hjr_scene_eval_camera(const hjr_scene*, const hjr_render_option*, float time, hjr_camera* out);
/* Texture(LUT_path, NonColor) — renderer/texture.h:16-39, loader/texture_load.h:7-20: 8-bit RGBA, caller frees with hjr_free */
int hjr_load_png_rgba8(const char* path, uint8_t** rgba, int* w, int* h);
/* the material-texture form of the same loader: PNG or baseline JPEG by file signature (stbi_load decodes both) */
int hjr_load_image_rgba8(const char* path, uint8_t** rgba, int* w, int* h);
/* HDRTexture(filename, background) — renderer/texture.h:67-100 (stbi_loadf): Radiance .hdr (RGBE) -> float RGBA (a = 0), caller frees with hjr_free */
int hjr_load_hdr_rgba32f(const char* path, float** rgba, int* w, int* h);
void hjr_free(void*);

/* ---------------- device side: replaces context/GAS/IAS/pipeline/SBT + optixLaunch ---------------- */
int hjr_create(int device_ordinal, hjr_ctx** out);                 /* optixDeviceContextInitialize, renderer.h:293-312 */
void hjr_destroy(hjr_ctx*);
/* cpySceneDataToDevice + optixTraversalBuild(GAS) + optixSBTBuild — renderer.h:197-255, 314-396, 620-739 */
int hjr_upload_scene(hjr_ctx*, const hjr_scene_view*);
/* updateIASMatrix + buildIAS — renderer.h:257-291, 398-490.  Flattens to world space and (re)builds the BVH. */
int hjr_set_transforms(hjr_ctx*, const float* transforms12, const float* inv_transforms12, uint32_t n_instances);
/* The same in two halves for pipelined frame loops: hjr_prepare_transforms does the host work (flatten + BVH build, worker
 * threads) into a pending slot and touches neither the device nor anything a running render reads, so it may run on another
 * thread while the previous frame renders; hjr_commit_transforms uploads the pending data and makes it current (call it after
 * that render has finished).  Unchanged transforms (static geometry) are detected and cost nothing. */
int hjr_prepare_transforms(hjr_ctx*, const float* transforms12, const float* inv_transforms12, uint32_t n_instances);
int hjr_commit_transforms(hjr_ctx*);
/* setLUT — renderer.h:854-898 (uchar4, normalised float read, linear, wrap).  NULL clears. */
int hjr_set_lut(hjr_ctx*, const uint8_t* rgba, int w, int h);
/* setSky — renderer.h:802-851: equirect float4 IBL texture (wrap, linear, element read).  NULL restores the 1x1 texel
 * `hjr_params.sky` (scene_sky_default).  Direction -> (u, v): u = atan2(d.z, d.x) / 2pi + 0.5, v = acos(d.y) / pi (build-defined). */
int hjr_set_sky(hjr_ctx*, const float* rgba32f, int w, int h);
/* Params fill + optixLaunch + CUDA_SYNC_CHECK + AOV D->H — renderer.h:1175-1242, 103-136.
 * Host buffers, width*height*4 floats each (albedo/normal may be NULL).  Synchronous. */
int hjr_render(hjr_ctx*, const hjr_params*, float* aov_color, float* aov_albedo, float* aov_normal);
/* Same launch writing float4 AOVs straight into caller-owned DEVICE memory (e.g. a torch tensor that then goes
 * through an RCCL collective), enqueued on `hip_stream` (hipStream_t as void*, NULL = default stream).  Asynchronous. */
int hjr_render_device(hjr_ctx*, const hjr_params*, void* d_aov_color, void* d_aov_albedo, void* d_aov_normal,
                      void* hip_stream);
int hjr_synchronize(hjr_ctx*);
/* ---- pixel-tile shard helpers (no reference counterpart: the reference is single-GPU, renderer.h:1077-1078) ----
 * Tiles are 8x8 pixels; tile (tx, ty) has id t = ty * tiles_x + (tx + ty) % tiles_x (row ty of tiles rotated by ty places, so that a
 * rank's tiles run along diagonals instead of forming vertical stripes when tiles_x is a multiple of world_size); tile t belongs to
 * rank t % world_size and is that rank's (t / world_size)-th tile. */
uint32_t hjr_owned_tiles(uint32_t width, uint32_t height, uint32_t rank, uint32_t world_size);
/* host arrays: row-major float4 frame <-> packed [owned tile][64] float4 of one rank (frame pixels of other ranks untouched) */
int hjr_pack_tiles(const float* frame_rgba, uint32_t width, uint32_t height, uint32_t rank, uint32_t world_size, float* packed_rgba);
int hjr_unpack_tiles(const float* packed_rgba, uint32_t width, uint32_t height, uint32_t rank, uint32_t world_size, float* frame_rgba);
/* the same on device pointers, asynchronous on `hip_stream` (NULL = the context's stream) */
int hjr_pack_tiles_device(hjr_ctx*, const void* d_frame, uint32_t width, uint32_t height, uint32_t rank, uint32_t world_size, void* d_packed, void* hip_stream);
int hjr_unpack_tiles_device(hjr_ctx*, const void* d_packed, uint32_t width, uint32_t height, uint32_t rank, uint32_t world_size, void* d_frame, void* hip_stream);
/* OptixDenoiserManager::layerSet + denoise() — renderer/denoiser.h:42-189, renderer/renderer.h:1093-1120, 1258-1270:
 * (aov_color | guide albedo | guide normal) of in_w x in_h -> AOV_Output of out_w x out_h.  The OptiX AI network is closed, so
 * this is a REPLACEMENT with the same data flow, not a reproduction of its pixels: HJR_MODE_DEFAULT copies
 * (blendFactor 1), HJR_MODE_DENOISE runs a 5-pass edge-avoiding a-trous filter guided by the two AOVs (out size == in size),
 * HJR_MODE_DENOISE_UPSCALE2X filters at (out_w / 2, out_h / 2) (renderer.h:1096-1099) and upscales 2x bilinearly. */
int hjr_denoise(hjr_ctx*, int render_mode, uint32_t in_w, uint32_t in_h, const float* aov_color, const float* aov_albedo,
                const float* aov_normal, float* out, uint32_t out_w, uint32_t out_h);
/* One frame of the loop in any render mode, kept on the device: launch at p->width x p->height (the caller halves it for
 * DenoiseUpScale2X, renderer.h:1096-1099), hjr_denoise_device, download of AOV_Output only (renderer.h:1229-1281).  Synchronous. */
int hjr_render_denoised(hjr_ctx*, const hjr_params*, int render_mode, float* out, uint32_t out_w, uint32_t out_h);
/* the same on device pointers (float4 images), asynchronous on `hip_stream` (NULL = the context's stream) */
int hjr_denoise_device(hjr_ctx*, int render_mode, uint32_t in_w, uint32_t in_h, const void* d_color, const void* d_albedo,
                       const void* d_normal, void* d_out, uint32_t out_w, uint32_t out_h, void* hip_stream);
/* The 8-bit preview buffer of the raygen program — `uchar4* image` of Params, allocated at renderer/renderer.h:1102 and bound at :1175, written
 * by the missing __raygen__rg and never read back by the host.  Build-defined: a float4 colour image on the device -> tonemapper of kernel/color.h
 * (HJR_TONEMAP_*) -> toSRGB + quantise (renderer.h:73-101) -> width*height RGBA8 on the device; asynchronous on `hip_stream` (NULL = the
 * context's stream).  The host form is hjr_tonemap_to_srgb8; device pow / exp may differ from libm by one code value at a quantisation boundary. */
int hjr_preview_device(hjr_ctx*, const void* d_color, uint32_t width, uint32_t height, int tonemap, void* d_rgba8, void* hip_stream);
int hjr_get_stats(hjr_ctx*, hjr_stats* out);
/* Tuning / test options of a context.  The library reads NO environment variable: kernel selection and layouts depend on the scene, the
 * launch parameters and these options only.  value -1 = the library's default; HJR_ERR_ARG for an unknown key or a value out of range.
 *   key               values      meaning (default)
 *   "pipeline"        0 1 2       kernel family: 0 per launch what was measured faster (wavefront kernels for MIS, megakernel otherwise; default),
 *                                 1 persistent megakernel, 2 workgroup-local wavefront kernel
 *   "lds_bvh"         0 1         1: stage BVH2 + triangles in LDS when they fit (default), 0: always read the scene from memory     [*]
 *   "lds_stack16"     0 1         1: 16-bit LDS traversal-stack entries whenever the tree admits them (default: only when 32-bit ones do not fit) [*]
 *   "bvh_width"       2 4         force the node format; 4 also forces the memory path (default: BVH2 in LDS when it fits, BVH4 otherwise)   [*]
 *   "leaf_max"        1..4        triangles per BVH leaf (2)                                                                        [*]
 *   "bvh_refine"      0..16       insertion-based refinement passes over the built BVH2, largest subtrees first (0 up to 65536 triangles, 1 above) [*]
 *   "node_min"        1..64       traversal descent loops: lanes still descending below which a pass moves on to the leaves (6 / 8 / 24 by layout and family)
 *   "hold_min"        0..64       megakernel, LDS layouts: metallic hits a wave collects before it shades them; 0 never holds (8)
 *   "hold_age"        1..1000     ... or rounds the oldest of them has waited (2)
 *   "short_stack"     1..64       memory layouts: traversal-stack entries per lane kept in LDS, deeper ones overflow to HBM (16)
 *   "blocks_per_cu"   1..8        memory layouts: workgroups per CU of the persistent grid (occupancy query)
 *   "top_nodes"       0..1024     memory layouts, BVH4: nodes of the top of the tree every workgroup also keeps in LDS (85 = levels 0 - 3; 0 none)
 *   "tile_order"      0 1 2       0 plain tile order, 1 first-hit classes, 2 classes + measured cost of the previous frame (1 on one GPU, 2 when sharded)
 *   "wf_cap"          64..32768   wavefront kernel: path contexts per workgroup, a power of two (2048 in LDS layouts, 4096 otherwise)
 *   "wf_refill" / "wf_prefetch_min" / "wf_trace_min"   wavefront kernel: hand-over thresholds of the trace stage, scheduler preference
 *   "host_threads"    1..256      worker threads of the per-frame host preparation, process-wide (min(hardware threads, 16))
 *   "verbose"         0 1         BVH format, sizes, host build stages per frame on stderr (0)
 *   "force_rebuild"   0 1         rebuild the frame data even when the transforms did not change (0)
 *   [*] takes effect at the next hjr_set_transforms / hjr_prepare_transforms.
 * No reference counterpart (OptiX owns these decisions); tests use them to force every kernel layout. */
int hjr_set_option(hjr_ctx*, const char* key, int value);
int hjr_get_option(hjr_ctx*, const char* key, int* value);
/* Host-only self-test of the 16-bit traversal-stack encoding (csrc/hjr_traverse.hip.h): 0 when every child ref of a tree the
 * builder admits to that layout survives encode + decode.  No reference counterpart (OptiX owns its traversal stack). */
int hjr_selftest_stack16(void);

/* ---------------- output stage (host) ---------------- */
/* float4ConvertColor: toSRGB + quantizeUnsignedChar — renderer/renderer.h:73-101 */
int hjr_float4_to_srgb8(const float* rgba, uint8_t* out_rgba8, uint32_t n_pixels);
/* Preview-buffer tonemappers of kernel/color.h: HJR_TONEMAP_UCHIMURA (color.h:10-53), HJR_TONEMAP_ACES (color.h:55-63),
 * applied per channel before the sRGB + quantise stage above (the raygen code that used them is missing: build-defined order). */
enum { HJR_TONEMAP_NONE = 0, HJR_TONEMAP_UCHIMURA = 1, HJR_TONEMAP_ACES = 2 };
int hjr_tonemap_to_srgb8(const float* rgba, uint8_t* out_rgba8, uint32_t n_pixels, int tonemap);
/* sutil::saveImage(name, buffer, false) — renderer.h:1291-1302.  flip_y != 0 writes row 0 at the bottom. */
int hjr_write_png(const char* path, const uint8_t* rgba8, uint32_t width, uint32_t height, int flip_y);
int hjr_write_pfm(const char* path, const float* rgba, uint32_t width, uint32_t height);
/* Renderer::initializeAndRender(render_option_path) — renderer/renderer.h:1053-1317, whole file-to-PNG path */
int hjr_render_file(const char* render_option_json, int device_ordinal);

#ifdef __cplusplus
}
#endif
#endif
