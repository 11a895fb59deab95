// Device context of libhenjou_hip.so: HBM-resident scene, per-frame BVH upload, kernel launch and timing.
// Replaces the reference's CUDA/OptiX plumbing (renderer/renderer.h:197-255 upload, 293-739 context/GAS/IAS/pipeline/SBT,
// 1175-1242 Params fill + optixLaunch).  No CPU fallback exists: without a gfx950 device every entry point fails loudly.
#include <hip/hip_runtime.h>
#include <chrono>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/henjou_hip.h"
#include "../host/frame.hpp"
#include "hjr_launch.hip.h"
#include "../host/abi.hpp"
#include "hjr_aux.hip.h"
#include "hjr_denoise.hip.h"

#define HIPCHK(call)                                                                                            \
    do {                                                                                                        \
        hipError_t e_ = (call);                                                                                 \
        if (e_ != hipSuccess) {                                                                                 \
            set_error(std::string(#call) + " failed: " + hipGetErrorString(e_));                               \
            return HJR_ERR_DEVICE;                                                                              \
        }                                                                                                       \
    } while (0)

extern "C" int hjr_create(int device, hjr_ctx** out)
{
    if (!out) { set_error("hjr_create: null out pointer"); return HJR_ERR_ARG; }
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_error(std::string("hjr_create: no HIP device available (") + hipGetErrorString(e) + "); this library has no CPU fallback");
        return HJR_ERR_DEVICE;
    }
    if (device < 0 || device >= n) { set_error("hjr_create: device ordinal out of range"); return HJR_ERR_ARG; }
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error(std::string("hjr_create: device is ") + prop.gcnArchName + ", this library carries gfx950 (MI355X) code objects only");
        return HJR_ERR_DEVICE;
    }
    hjr_ctx* c = new hjr_ctx();
    c->device = device;
    c->n_cus = prop.multiProcessorCount;
    memset(&c->stats, 0, sizeof(c->stats));
#ifdef HJR_ENV_OPTIONS /* experiment builds only (make variant): every option also from the environment, HJR_<KEY> */
    for (int i = 0; i < hjr::OPT_COUNT; i++) {
        std::string name = std::string("HJR_") + hjr::opt_table()[i].key;
        for (char& ch : name) ch = (char)toupper((unsigned char)ch);
        if (const char* e = getenv(name.c_str())) { const int v = atoi(e); if (v >= hjr::opt_table()[i].lo && v <= hjr::opt_table()[i].hi) c->opt.v[i] = v; }
    }
    if (c->opt.is_set(hjr::OPT_HOST_THREADS)) hjr::set_host_threads(c->opt.v[hjr::OPT_HOST_THREADS]);
#endif
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess || hipEventCreate(&c->ev0) != hipSuccess ||
        hipEventCreate(&c->ev1) != hipSuccess) {
        set_error("hjr_create: stream/event creation failed");
        delete c;
        return HJR_ERR_DEVICE;
    }
    *out = c;
    return HJR_OK;
}

// Tuning / test options (host/options.hpp lists keys, ranges and defaults; include/henjou_hip.h documents them)
extern "C" int hjr_set_option(hjr_ctx* c, const char* key, int value)
{
    if (!c) { set_error("hjr_set_option: null context"); return HJR_ERR_ARG; }
    const int i = hjr::opt_find(key);
    if (i < 0) { set_error(std::string("hjr_set_option: unknown option \"") + (key ? key : "(null)") + "\""); return HJR_ERR_ARG; }
    const hjr::OptDesc& d = hjr::opt_table()[i];
    if (value != -1 && (value < d.lo || value > d.hi || (i == hjr::OPT_BVH_WIDTH && value == 3) || (i == hjr::OPT_WF_CAP && (value & (value - 1)) != 0))) {
        set_error(std::string("hjr_set_option: value out of range for \"") + d.key + "\" (" + std::to_string(d.lo) + " .. " + std::to_string(d.hi) + ", or -1 for the default)");
        return HJR_ERR_ARG;
    }
    c->opt.v[i] = value;
    if (i == hjr::OPT_HOST_THREADS) hjr::set_host_threads(value);
    return HJR_OK;
}
extern "C" int hjr_get_option(hjr_ctx* c, const char* key, int* value)
{
    if (!c || !value) { set_error("hjr_get_option: null argument"); return HJR_ERR_ARG; }
    const int i = hjr::opt_find(key);
    if (i < 0) { set_error(std::string("hjr_get_option: unknown option \"") + (key ? key : "(null)") + "\""); return HJR_ERR_ARG; }
    *value = c->opt.v[i];
    return HJR_OK;
}

extern "C" void hjr_destroy(hjr_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    for (DevBuf* b : { &c->d_nodes, &c->d_tri_geom, &c->d_tri_shade, &c->d_tri_inst, &c->d_materials, &c->d_lights, &c->d_lut, &c->d_spill, &c->d_wf_ctx, &c->d_tiles, &c->d_tile_cost, &c->d_dn_a, &c->d_dn_b, &c->d_dn_out,
                       &c->d_texels, &c->d_tex_desc, &c->d_srgb_lut, &c->d_sky, &c->d_work, &c->d_color, &c->d_albedo, &c->d_normal, &c->d_part_color, &c->d_part_albedo, &c->d_part_normal })
        b->release();
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int hjr_upload_scene(hjr_ctx* c, const hjr_scene_view* v)
{
    if (!c || !v) { set_error("hjr_upload_scene: null argument"); return HJR_ERR_ARG; }
    hjr_scene_view view; // sized struct: only the bytes the caller owns are read
    if (!hjr::abi_take(v, view, "hjr_upload_scene")) return HJR_ERR_ARG;
    v = &view;
    std::string err;
    if (!c->scene.set(*v, err)) { set_error("hjr_upload_scene: " + err); return HJR_ERR_ARG; }
    HIPCHK(hipSetDevice(c->device));
    static_assert(sizeof(hjr_material) == HJR_MAT_F4 * 16, "hjr_material must be HJR_MAT_F4 x float4");
    if (!c->d_materials.upload(c->scene.materials.data(), c->scene.materials.size() * sizeof(hjr_material), c->stream)) {
        set_error("hjr_upload_scene: material upload failed");
        return HJR_ERR_DEVICE;
    }
    // textureBind (renderer.h:740-800): RGBA8 atlas + per-slot descriptors + the sRGB decode table
    c->n_textures = (uint32_t)c->scene.textures.size();
    if (c->n_textures) {
        std::vector<uint32_t> desc;
        for (auto& t : c->scene.textures) { desc.push_back(t.offset); desc.push_back(t.width); desc.push_back(t.height); desc.push_back((uint32_t)t.srgb); }
        float lut[256];
        for (int i = 0; i < 256; i++) {
            double v = (double)i / 255.0;
            lut[i] = (float)(v <= 0.04045 ? v / 12.92 : pow((v + 0.055) / 1.055, 2.4));
        }
        if (!c->d_texels.upload(c->scene.texels.data(), c->scene.texels.size() * 4, c->stream) ||
            !c->d_tex_desc.upload(desc.data(), desc.size() * 4, c->stream) || !c->d_srgb_lut.upload(lut, sizeof(lut), c->stream)) {
            set_error("hjr_upload_scene: texture upload failed");
            return HJR_ERR_DEVICE;
        }
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    c->have_scene = true;
    c->have_frame = false;
    return HJR_OK;
}

// updateIASMatrix + buildIAS (renderer.h:257-291, 398-490) in two halves, so that a frame loop can prepare frame f + 1 on the
// host (worker threads, no device access, no access to what a running render reads) while frame f renders:
//   hjr_prepare_transforms: flatten + BVH build into the context's PENDING frame data;
//   hjr_commit_transforms:  upload the pending data (after the previous render has finished) and make it current.
// hjr_set_transforms = prepare + commit.
extern "C" int hjr_prepare_transforms(hjr_ctx* c, const float* m, const float* inv, uint32_t n)
{
    if (!c || (n && (!m || !inv))) { set_error("hjr_set_transforms: null argument"); return HJR_ERR_ARG; }
    if (!c->have_scene) { set_error("hjr_set_transforms: no scene uploaded"); return HJR_ERR_STATE; }
    std::string err;
    hjr::BuildOptions bo; // options "lds_bvh", "lds_stack16", "bvh_width", "leaf_max", "verbose" (host build stages)
    bo.allow_lds = c->opt.get(hjr::OPT_LDS_BVH, 1) != 0;
    bo.prefer_stack16 = c->opt.get(hjr::OPT_LDS_STACK16, 0) != 0;
    bo.bvh_width = c->opt.get(hjr::OPT_BVH_WIDTH, -1);
    bo.leaf_max = c->opt.get(hjr::OPT_LEAF_MAX, -1);
    bo.refine = c->opt.get(hjr::OPT_BVH_REFINE, -1);
    bo.timing = c->opt.get(hjr::OPT_VERBOSE, 0) != 0;
    const uint32_t build_tag = (bo.allow_lds ? 1u : 0u) | (bo.prefer_stack16 ? 2u : 0u) | ((uint32_t)(bo.bvh_width + 1) << 2) | ((uint32_t)(bo.leaf_max + 1) << 6) | ((uint32_t)(bo.refine + 1) << 10);
    c->pending_valid = false;
    c->pending_same = false;
    // unchanged instance transforms (static geometry, e.g. a camera-only animation): the world-space arrays and the BVH of the
    // previous frame are still right; the reference re-uploads its IAS every frame (renderer.h:257-291), which costs it nothing
    const bool force_rebuild = c->opt.get(hjr::OPT_FORCE_REBUILD, 0) != 0; // benchmarking option
    if (!force_rebuild && c->have_frame && c->last_build_tag == build_tag && c->last_m.size() == (size_t)n * 12 && n == c->scene.n_instances &&
        (n == 0 || (memcmp(c->last_m.data(), m, (size_t)n * 48) == 0 && memcmp(c->last_inv.data(), inv, (size_t)n * 48) == 0))) {
        c->pending_same = true;
        c->pending_valid = true;
        return HJR_OK;
    }
    const auto t_build0 = std::chrono::steady_clock::now();
    if (!hjr::build_frame(c->scene, m, inv, n, bo, c->pending, err)) { set_error("hjr_set_transforms: " + err); return HJR_ERR_ARG; }
    c->pending_build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_build0).count();
    c->pending_m.assign(m, m + (size_t)n * 12); c->pending_inv.assign(inv, inv + (size_t)n * 12); c->pending_build_tag = build_tag;
    c->pending_valid = true;
    return HJR_OK;
}

extern "C" int hjr_commit_transforms(hjr_ctx* c)
{
    if (!c) { set_error("hjr_commit_transforms: null context"); return HJR_ERR_ARG; }
    if (!c->pending_valid) { set_error("hjr_commit_transforms: nothing prepared"); return HJR_ERR_STATE; }
    c->pending_valid = false;
    if (c->pending_same) {
        if (c->opt.get(hjr::OPT_VERBOSE, 0)) fprintf(stderr, "[hjr] transforms unchanged: frame data reused\n");
        return HJR_OK;
    }
    HIPCHK(hipSetDevice(c->device));
    std::swap(c->frame, c->pending);
    const hjr::FrameData& f = c->frame;
    bool ok = c->d_nodes.upload(f.nodes.data(), f.nodes.size() * 4, c->stream) &&
              c->d_tri_geom.upload(f.tri_geom.data(), f.tri_geom.size() * 4, c->stream) &&
              c->d_tri_shade.upload(f.tri_shade.data(), f.tri_shade.size() * 4, c->stream) &&
              c->d_tri_inst.upload(f.tri_inst.data(), f.tri_inst.size() * 4, c->stream) &&
              c->d_lights.upload(f.lights.data(), f.lights.size() * 4, c->stream);
    if (!ok) { c->have_frame = false; set_error("hjr_set_transforms: device upload failed"); return HJR_ERR_DEVICE; }
    HIPCHK(hipStreamSynchronize(c->stream));
    c->have_frame = true;
    c->last_m.swap(c->pending_m); c->last_inv.swap(c->pending_inv); c->last_build_tag = c->pending_build_tag;
    c->stats.bvh_nodes = f.n_nodes;
    c->stats.bvh_depth = f.depth;
    if (c->opt.get(hjr::OPT_VERBOSE, 0)) fprintf(stderr, "[hjr] BVH%u (lds_mode %d): %u nodes (%zu KB), %u triangles (%zu KB), stack %u entries/lane, host build %.1f ms\n", f.width, f.lds_mode, f.n_nodes, f.nodes.size() * 4 / 1024, f.n_tris, f.tri_geom.size() * 4 / 1024, f.stack_need, c->pending_build_ms);
    c->stats.n_triangles = f.n_tris;
    return HJR_OK;
}

extern "C" int hjr_set_transforms(hjr_ctx* c, const float* m, const float* inv, uint32_t n)
{
    const int rc = hjr_prepare_transforms(c, m, inv, n);
    return rc != HJR_OK ? rc : hjr_commit_transforms(c);
}

extern "C" int hjr_set_lut(hjr_ctx* c, const uint8_t* rgba, int w, int h)
{
    if (!c) { set_error("hjr_set_lut: null context"); return HJR_ERR_ARG; }
    HIPCHK(hipSetDevice(c->device));
    if (!rgba || w <= 0 || h <= 0) { c->lut_w = c->lut_h = 0; return HJR_OK; }
    if (!c->d_lut.upload(rgba, (size_t)w * (size_t)h * 4, c->stream)) { set_error("hjr_set_lut: upload failed"); return HJR_ERR_DEVICE; }
    HIPCHK(hipStreamSynchronize(c->stream));
    c->lut_w = w; c->lut_h = h;
    return HJR_OK;
}


// persistent grid = resident workgroups only: CUs x (workgroups the kernel's VGPR/LDS budget admits per CU), capped by the
// number of wavefront-sized batches of work; HJR_BLOCKS_PER_CU overrides the occupancy query
extern "C" int hjr_set_sky(hjr_ctx* c, const float* rgba, int w, int h)
{
    if (!c) { set_error("hjr_set_sky: null context"); return HJR_ERR_ARG; }
    HIPCHK(hipSetDevice(c->device));
    if (!rgba || w <= 0 || h <= 0) { c->sky_w = c->sky_h = 0; return HJR_OK; }
    if (!c->d_sky.upload(rgba, (size_t)w * (size_t)h * 16, c->stream)) { set_error("hjr_set_sky: upload failed"); return HJR_ERR_DEVICE; }
    HIPCHK(hipStreamSynchronize(c->stream));
    c->sky_w = w; c->sky_h = h;
    return HJR_OK;
}

// the render kernels live in their own translation units (hjr_launch.hip.h)
#ifndef HJR_LEAN_VARIANT
extern template int hjr_launch<HJR_INTEGRATOR_NEE, false>(hjr_ctx*, const KParams&, uint64_t, int, hipStream_t);
extern template int hjr_launch<HJR_INTEGRATOR_NEE, true>(hjr_ctx*, const KParams&, uint64_t, int, hipStream_t);
extern template int hjr_launch<HJR_INTEGRATOR_PT, false>(hjr_ctx*, const KParams&, uint64_t, int, hipStream_t);
extern template int hjr_launch<HJR_INTEGRATOR_PT, true>(hjr_ctx*, const KParams&, uint64_t, int, hipStream_t);
extern template int hjr_launch<HJR_INTEGRATOR_MIS, false>(hjr_ctx*, const KParams&, uint64_t, int, hipStream_t);
extern template int hjr_launch<HJR_INTEGRATOR_MIS, true>(hjr_ctx*, const KParams&, uint64_t, int, hipStream_t);
#endif
#if !defined(HJR_LEAN_VARIANT) && !defined(HJR_UNITY)
template <int I> int hjr_launch_fast(hjr_ctx*, const KParams&, uint64_t, int, hipStream_t); // hjr_launch_fast_*.hip (HJR_FLAG_FAST_MATH)
extern template int hjr_launch_fast<HJR_INTEGRATOR_NEE>(hjr_ctx*, const KParams&, uint64_t, int, hipStream_t);
extern template int hjr_launch_fast<HJR_INTEGRATOR_PT>(hjr_ctx*, const KParams&, uint64_t, int, hipStream_t);
#define HJR_HAVE_FAST 1
#endif
#ifdef HJR_UNITY /* diagnostic variants (make variant): one translation unit, so that the __device__ diagnostic counters are one symbol */
#ifdef HJR_LEAN_VARIANT /* NEE without the statistics counters only; every other launch runs that kernel too (timing experiments, not pictures) */
#include "hjr_launch.hip.h"
template int hjr_launch<HJR_INTEGRATOR_NEE, false>(hjr_ctx*, const KParams&, uint64_t, int, hipStream_t);
#else
#include "hjr_launch_nee.hip"
#include "hjr_launch_pt.hip"
#include "hjr_launch_mis.hip"
#endif
#endif

static int render_impl(hjr_ctx* c, const hjr_params* p, void* d_color, void* d_albedo, void* d_normal, hipStream_t st)
{
    if (!c || !p || !d_color) { set_error("hjr_render: null argument"); return HJR_ERR_ARG; }
    if (!c->have_scene || !c->have_frame) { set_error("hjr_render: upload a scene and set transforms first"); return HJR_ERR_STATE; }
    if (p->width == 0 || p->height == 0 || p->spp == 0) { set_error("hjr_render: width, height and spp must be positive"); return HJR_ERR_ARG; }
    if (p->width > 8192 || p->height > 8192) { set_error("hjr_render: frames larger than 8192 x 8192 are not supported"); return HJR_ERR_ARG; }
    if (p->integrator > HJR_INTEGRATOR_MIS) { set_error("hjr_render: unknown integrator"); return HJR_ERR_ARG; }
    const uint32_t world = p->world_size ? p->world_size : 1u;
    if (p->rank >= world) { set_error("hjr_render: rank >= world_size"); return HJR_ERR_ARG; }
    HIPCHK(hipSetDevice(c->device));

    const uint32_t tiles_x = (p->width + HJR_TILE - 1) / HJR_TILE, tiles_y = (p->height + HJR_TILE - 1) / HJR_TILE;
    const uint64_t n_tiles = (uint64_t)tiles_x * tiles_y;
    const uint64_t owned = (n_tiles > p->rank) ? (n_tiles - p->rank + world - 1) / world : 0;
    const uint32_t chunk_spp = hjr_chunk_spp(p->spp), n_chunks = hjr_n_chunks(p->spp);
    const uint64_t n_items = owned * n_chunks * 64;
    // the 32-bit queue head overshoots n_items by at most 64 per wave of the persistent grid (every wave stops fetching once it
    // has seen the queue dry, hjr_kernel.hip.h); 2^24 covers 262 144 waves, far more than any resident grid
    if (n_items >= 0xffffffffull - (1ull << 24)) { set_error("hjr_render: image too large (more than 2^32 - 2^24 work items per launch)"); return HJR_ERR_ARG; }

    // work area: [0] queue head, [16..] HJR_NSTAT uint64 counters
    const size_t nan_list_at = 16 + (HJR_NSTAT + 20) * 8 + 32 + 512;
    const size_t work_bytes = nan_list_at + (1 + HJR_NAN_LIST) * 8; // +20: phase clocks / lane-occupancy sums of the HJR_TIMING diagnostic build; +32: tile-class counters
    if (c->d_work.cap < work_bytes) {
        std::vector<unsigned char> z(work_bytes, 0);
        if (!c->d_work.upload(z.data(), work_bytes, st)) { set_error("hjr_render: work buffer allocation failed"); return HJR_ERR_DEVICE; }
    }
    HIPCHK(hipMemsetAsync(c->d_work.p, 0, work_bytes, st));
    const size_t img_bytes = (size_t)p->width * p->height * 16;
    const bool packed = (p->flags & HJR_FLAG_PACKED) != 0;
    if (!packed && world > 1 && (p->flags & HJR_FLAG_ZERO_UNOWNED)) {
        HIPCHK(hipMemsetAsync(d_color, 0, img_bytes, st));
        if (d_albedo) HIPCHK(hipMemsetAsync(d_albedo, 0, img_bytes, st));
        if (d_normal) HIPCHK(hipMemsetAsync(d_normal, 0, img_bytes, st));
    }

    KParams kp;
    memset(&kp, 0, sizeof(kp));
    kp.n_owned_tiles = (uint32_t)owned;
    if (n_chunks > 1) {
        // chunk sums of THIS rank's tiles only: [chunk][owned tile][64] float4 (1/world of the frame; allocated once per size)
        const size_t part_bytes = (size_t)owned * 64u * 16u * n_chunks;
        DevBuf* pb[3] = { &c->d_part_color, &c->d_part_albedo, &c->d_part_normal };
        void* want[3] = { d_color, d_albedo, d_normal };
        for (int i = 0; i < 3; i++) {
            if (!want[i]) continue;
            if (pb[i]->cap < part_bytes) {
                pb[i]->release();
                if (hipMalloc(&pb[i]->p, part_bytes) != hipSuccess) { set_error("hjr_render: chunk-sum buffer allocation failed"); return HJR_ERR_DEVICE; }
                pb[i]->cap = part_bytes;
            }
        }
        kp.part_color = (float4*)c->d_part_color.p;
        kp.part_albedo = d_albedo ? (float4*)c->d_part_albedo.p : nullptr;
        kp.part_normal = d_normal ? (float4*)c->d_part_normal.p : nullptr;
    }
    kp.chunk_spp = chunk_spp; kp.n_chunks = n_chunks;
    kp.nodes = (const float4*)c->d_nodes.p;
    kp.tri_geom = (const float4*)c->d_tri_geom.p;
    kp.tri_shade = (const float4*)c->d_tri_shade.p;
    kp.tri_inst = (const uint32_t*)c->d_tri_inst.p;
    kp.materials = (const float4*)c->d_materials.p;
    kp.lights = (const float4*)c->d_lights.p;
    kp.lut = (c->lut_w > 0) ? (const uchar4*)c->d_lut.p : nullptr;
    kp.lut_w = c->lut_w; kp.lut_h = c->lut_h;
    if (c->n_textures) { kp.texels = (const uchar4*)c->d_texels.p; kp.tex_desc = (const uint4*)c->d_tex_desc.p; kp.srgb_lut = (const float*)c->d_srgb_lut.p; }
    if (c->sky_w > 0) { kp.sky_tex = (const float4*)c->d_sky.p; kp.sky_w = c->sky_w; kp.sky_h = c->sky_h; }
    kp.ibl_intensity = p->ibl_intensity;
    kp.aov_color = (float4*)d_color; kp.aov_albedo = (float4*)d_albedo; kp.aov_normal = (float4*)d_normal;
    kp.queue_head = (unsigned int*)c->d_work.p;
    kp.stats = (unsigned long long*)((char*)c->d_work.p + 16);
    kp.nan_list = (unsigned long long*)((char*)c->d_work.p + nan_list_at);
    kp.n_lights = c->frame.n_lights;
    kp.width = p->width; kp.height = p->height; kp.spp = p->spp; kp.frame = p->frame; kp.seed = p->seed; kp.integrator = p->integrator;
    kp.tiles_x = tiles_x; kp.n_owned_items = (uint32_t)n_items;
    kp.rank = p->rank; kp.world = world;
    kp.packed = packed ? 1u : 0u;
    for (int k = 0; k < 3; k++) {
        kp.cam_pos[k] = p->camera.pos[k]; kp.cam_dir[k] = p->camera.dir[k];
        kp.cam_up[k] = p->camera.up[k]; kp.cam_right[k] = p->camera.right[k];
        kp.sky[k] = p->sky[k] * p->ibl_intensity; // __miss__ms: texel * params.ibl_intensity
    }
    kp.cam_f = p->camera.f;

    const bool stats = (p->flags & HJR_FLAG_STATS) != 0;
    kp.n_node_f4 = c->frame.n_nodes * (c->frame.width == 2 ? HJR_NODE2_F4 : HJR_NODE4_F4);
    kp.n_tri_f4 = (c->frame.n_tris ? c->frame.n_tris : 1u) * HJR_TRI_F4;
    kp.stack_depth = c->frame.stack_need; // exact worst case for this tree (host/frame.cpp)
    kp.n_mat_f4 = (uint32_t)c->scene.materials.size() * HJR_MAT_F4;
    kp.n_light_f4 = c->frame.n_lights * HJR_LIGHT_F4;
    // node format / LDS staging were decided by the host builder for this frame (host/frame.cpp)
    int lds_mode = c->frame.lds_mode;
    if (lds_mode == 0 && c->frame.width == 2) lds_mode = 3;
    c->stats.lds_mode = (uint32_t)lds_mode;
    c->stats.stack_need = c->frame.stack_need;
    c->stats.stack_lds_entries = 0; // set by the memory-path launch
    HIPCHK(hipEventRecord(c->ev0, st));
    // cost-ordered tile list (hjr_classify_tiles_kernel): HJR_TILE_ORDER=0 keeps the plain round-robin order
    const int order_knob = c->opt.get(hjr::OPT_TILE_ORDER, -1); // option "tile_order"
    const bool tile_order_on = order_knob != 0;
    if (tile_order_on && owned > 0) {
        const size_t tb = (size_t)owned * 4;
        if (c->d_tiles.cap < 3 * tb) {
            c->d_tiles.release();
            if (hipMalloc(&c->d_tiles.p, 3 * tb) != hipSuccess) { set_error("hjr_render: tile list allocation failed"); return HJR_ERR_DEVICE; }
            c->d_tiles.cap = 3 * tb;
            c->cost_tag = 0; // the classes of the previous frames went with the buffer
        }
        kp.tile_order_w = (uint32_t*)c->d_tiles.p;
        kp.tile_class = (uint32_t*)((char*)c->d_tiles.p + tb);
        kp.tile_bucket = (uint32_t*)((char*)c->d_tiles.p + 2 * tb);
        kp.tile_count = (uint32_t*)((char*)c->d_work.p + 16 + (HJR_NSTAT + 20) * 8);
        // Inside a class the tiles can also be ordered by what they cost in the previous frame of the same configuration.  That
        // shortens the tail of a launch further (an 8-GPU share of C2: 19.0 -> 18.4 ms) but gives up the scanline order inside a
        // class, which costs 1.6 % when the launch is long (N = 1: 134.6 -> 136.8 ms): used when the frame is split over several
        // GPUs.  Pure scheduling: no pixel depends on it.  Option "tile_order" = 1 / 2 forces it off / on.
        const bool cost_feedback = order_knob == 2 || (order_knob != 1 && world > 1);
        const uint64_t tag = ((uint64_t)p->width << 48) ^ ((uint64_t)p->height << 32) ^ ((uint64_t)p->spp << 12) ^ ((uint64_t)world << 8) ^
                             ((uint64_t)p->rank << 2) ^ (uint64_t)p->integrator ^ 0x8000000000000000ull;
        bool have_cost = false;
        if (cost_feedback) {
            if (c->d_tile_cost.cap < tb) {
                c->d_tile_cost.release();
                if (hipMalloc(&c->d_tile_cost.p, tb) != hipSuccess) { set_error("hjr_render: tile cost allocation failed"); return HJR_ERR_DEVICE; }
                c->d_tile_cost.cap = tb;
                c->cost_tag = 0;
            }
            have_cost = c->cost_tag == tag;
            if (!have_cost) HIPCHK(hipMemsetAsync(c->d_tile_cost.p, 0, tb, st));
            c->cost_tag = tag;
            kp.tile_cost = (uint32_t*)c->d_tile_cost.p;
            kp.cost_hist = (uint32_t*)((char*)c->d_work.p + 16 + (HJR_NSTAT + 20) * 8 + 32);
            kp.cost_div = 64u * p->spp;
        }
        const unsigned tg = (unsigned)((owned + 255) / 256);
        if (have_cost) {
            hipLaunchKernelGGL(hjr_cost_hist_kernel, dim3(tg), dim3(256), 0, st, kp);
            hipLaunchKernelGGL(hjr_cost_scatter_kernel, dim3(tg), dim3(256), 0, st, kp);
        } else {
            const unsigned cg = (unsigned)std::min<uint64_t>(owned, (uint64_t)c->n_cus * 16);
            const size_t csm = (size_t)64 * kp.stack_depth * 4;
            if (c->frame.width == 2) hipLaunchKernelGGL(hjr_classify_tiles_kernel<2>, dim3(cg), dim3(64), csm, st, kp);
            else hipLaunchKernelGGL(hjr_classify_tiles_kernel<4>, dim3(cg), dim3(64), csm, st, kp);
            hipLaunchKernelGGL(hjr_order_tiles_kernel, dim3(tg), dim3(256), 0, st, kp);
        }
        HIPCHK(hipGetLastError());
        kp.tile_order = (const uint32_t*)c->d_tiles.p;
    }
    int lrc = 0;
    c->stats.fast_math = 0u;
#ifdef HJR_LEAN_VARIANT
    lrc = hjr_launch<HJR_INTEGRATOR_NEE, false>(c, kp, n_items, lds_mode, st);
#else
#ifdef HJR_HAVE_FAST
    // approximate-arithmetic kernels (megakernel family).  A counting launch stays exact, and so does MIS: its exact launch runs on the
    // wavefront kernels, which beat the approximate megakernel (C2: 175 vs 185 ms), so the flag would only make it slower
    if ((p->flags & HJR_FLAG_FAST_MATH) && !stats && p->integrator != HJR_INTEGRATOR_MIS) {
        c->stats.fast_math = 1u;
        lrc = p->integrator == HJR_INTEGRATOR_NEE ? hjr_launch_fast<HJR_INTEGRATOR_NEE>(c, kp, n_items, lds_mode, st) : hjr_launch_fast<HJR_INTEGRATOR_PT>(c, kp, n_items, lds_mode, st);
    } else
#endif
    switch (p->integrator * 2 + (stats ? 1 : 0)) {
    case 0: lrc = hjr_launch<HJR_INTEGRATOR_NEE, false>(c, kp, n_items, lds_mode, st); break;
    case 1: lrc = hjr_launch<HJR_INTEGRATOR_NEE, true>(c, kp, n_items, lds_mode, st); break;
    case 2: lrc = hjr_launch<HJR_INTEGRATOR_PT, false>(c, kp, n_items, lds_mode, st); break;
    case 3: lrc = hjr_launch<HJR_INTEGRATOR_PT, true>(c, kp, n_items, lds_mode, st); break;
    case 4: lrc = hjr_launch<HJR_INTEGRATOR_MIS, false>(c, kp, n_items, lds_mode, st); break;
    default: lrc = hjr_launch<HJR_INTEGRATOR_MIS, true>(c, kp, n_items, lds_mode, st); break;
    }
#endif
    if (lrc != 0) { set_error("hjr_render: could not reserve dynamic LDS for the BVH"); return HJR_ERR_DEVICE; }
    HIPCHK(hipGetLastError());
    if (n_chunks > 1) {
        const size_t n_slots = (size_t)owned * 64u;
        unsigned fb = (unsigned)std::max<size_t>(1, std::min<size_t>((n_slots + 255) / 256, (size_t)c->n_cus * 8));
        hipLaunchKernelGGL(hjr_finalize_kernel, dim3(fb), dim3(256), 0, st, kp);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipEventRecord(c->ev1, st));
    c->event_pending = true;
    return HJR_OK;
}

static int fetch_stats(hjr_ctx* c, hipStream_t st)
{
    unsigned long long h[HJR_NSTAT];
    HIPCHK(hipMemcpyAsync(h, (char*)c->d_work.p + 16, sizeof(h), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    uint64_t* dst = &c->stats.samples;
    for (int i = 0; i < 10; i++) dst[i] = h[i];
    c->stats.stack_overflow_pushes = h[10];
    {
        unsigned long long nl[1 + HJR_NAN_LIST];
        HIPCHK(hipMemcpy(nl, (char*)c->d_work.p + 16 + (HJR_NSTAT + 20) * 8 + 32 + 512, sizeof(nl), hipMemcpyDeviceToHost));
        const uint32_t n = (uint32_t)std::min<unsigned long long>(nl[0], HJR_NAN_LIST);
        c->stats.nan_located = n;
        for (uint32_t i = 0; i < HJR_NAN_LIST; i++) {
            const unsigned long long k = i < n ? nl[1 + i] : 0ull;
            c->stats.nan_where[i][0] = (uint32_t)(k & 0x1fffu); c->stats.nan_where[i][1] = (uint32_t)((k >> 13) & 0x1fffu); c->stats.nan_where[i][2] = (uint32_t)(k >> 26);
        }
    }
#ifdef HJR_WF_TIMING
    { // diagnostic build only: where the waves of the wavefront kernel spend their clocks
        unsigned long long d[24];
        (void)hipMemcpyFromSymbol(d, HIP_SYMBOL(wf_diag), sizeof(d));
        const double tot = (double)d[0] + (double)d[1] + (double)d[2];
        if (tot > 0) {
            fprintf(stderr, "[hjr wf timing] scheduler idle %.1f%%  trace stage %.1f%% (hand-overs %.1f%%)  shade stage %.1f%% (waiting for context loads %.1f%%, store + push %.1f%%)\n",
                    100 * d[0] / tot, 100 * d[1] / tot, 100 * d[9] / tot, 100 * d[2] / tot, 100 * d[3] / tot, 100 * d[10] / tot);
            fprintf(stderr, "[hjr wf timing] shade batches %llu, %.1f contexts each; trace calls %llu, hand-overs %llu with %.1f finished rays each\n", d[4], d[4] ? (double)d[5] / d[4] : 0.0,
                    d[8], d[6], d[6] ? (double)d[7] / d[6] : 0.0);
            fprintf(stderr, "[hjr wf timing] trace stage lanes: outer iterations %llu with %.1f lanes holding a ray; node steps %llu wave-iterations x %.1f lanes; triangle tests %llu x %.1f lanes\n",
                    d[15], d[15] ? (double)d[16] / d[15] : 0.0, d[11], d[11] ? (double)d[12] / d[11] : 0.0, d[13], d[13] ? (double)d[14] / d[13] : 0.0);
            unsigned long long z[24] = { 0 };
            (void)hipMemcpyToSymbol(HIP_SYMBOL(wf_diag), z, sizeof(z));
        }
    }
#endif
#ifdef HJR_WF_WATCHDOG
    { // diagnostic build only: did the wavefront kernel run into its deadline, and where?
        unsigned long long wd[19];
        HIPCHK(hipMemcpy(wd, (char*)c->d_work.p + 16 + HJR_NSTAT * 8, sizeof(wd), hipMemcpyDeviceToHost));
        unsigned int where[8] = { 0 };
        (void)hipMemcpyFromSymbol(where, HIP_SYMBOL(wf_where), sizeof(where));
        if (where[1] | where[2] | where[3] | where[4] | where[5] | where[6]) {
            fprintf(stderr, "[hjr wf watchdog] deadline hit in (lane counts): take %u, push slot-wait %u, push publish-wait %u, trace loop %u, pop %u, scheduler %u\n", where[1], where[2], where[3], where[4], where[5], where[6]);
            fprintf(stderr, "[hjr wf watchdog] first workgroup to give up: block %llu live %llu items_held %llu\n", wd[17], wd[16], wd[18]);
            for (int q = 0; q < 5; q++) fprintf(stderr, "   queue %d: commit %llu head %llu tail %llu\n", q, wd[1 + q], wd[6 + q], wd[11 + q]);
            unsigned int zero[8] = { 0 };
            (void)hipMemcpyToSymbol(HIP_SYMBOL(wf_where), zero, sizeof(zero));
        }
    }
#endif
#ifdef HJR_TIMING
    { // diagnostic build only: wave-clock shares of the megakernel's loop phases and lane occupancies
        unsigned long long tk[18];
        HIPCHK(hipMemcpy(tk, (char*)c->d_work.p + 16 + HJR_NSTAT * 8, sizeof(tk), hipMemcpyDeviceToHost));
        const double tot = (double)tk[0] + (double)tk[1] + (double)tk[2];
        if (tot > 0) fprintf(stderr, "[hjr timing] roulette/refill/regeneration %.1f%%  fused trace %.1f%%  resolve + hit program + shading %.1f%%  (%.3g wave-clocks)\n",
                             100 * tk[0] / tot, 100 * tk[1] / tot, 100 * tk[2] / tot, tot);
        if (tk[3]) fprintf(stderr, "[hjr timing] lanes per round: closest-hit ray %.1f, shadow ray %.1f, serviced %.1f\n", (double)tk[4] / tk[3], (double)tk[5] / tk[3], (double)tk[6] / tk[3]);
        const unsigned long long* td = tk + 7;
        if (td[0]) fprintf(stderr, "[hjr timing] traversal: %.1f passes per round, %.1f lanes with a ray per pass (%.1f on a shadow ray); node steps %.2f wave-iterations per pass x %.1f lanes; "
                                   "leaf parts in %.0f%% of the passes x %.1f lanes; triangle tests %.2f wave-iterations per pass x %.1f lanes\n",
                           tk[3] ? (double)td[0] / tk[3] : 0.0, (double)td[1] / td[0], (double)td[8] / td[0], (double)td[2] / td[0], td[2] ? (double)td[3] / td[2] : 0.0,
                           100.0 * td[4] / td[0], td[4] ? (double)td[5] / td[4] : 0.0, (double)td[6] / td[0], td[6] ? (double)td[7] / td[6] : 0.0);
    }
#endif
    return HJR_OK;
}

extern "C" int hjr_render_device(hjr_ctx* c, const hjr_params* p_user, void* d_color, void* d_albedo, void* d_normal, void* stream)
{
    hjr_params params; // sized struct
    if (!c || !hjr::abi_take(p_user, params, "hjr_render_device")) { if (!c) set_error("hjr_render_device: null context"); return HJR_ERR_ARG; }
    const hjr_params* p = &params;
    if (!c) { set_error("hjr_render_device: null context"); return HJR_ERR_ARG; }
    hipStream_t st = stream ? (hipStream_t)stream : c->stream;
    return render_impl(c, p, d_color, d_albedo, d_normal, st);
}

static bool ensure(DevBuf& b, size_t bytes)
{
    if (b.cap >= bytes) return true;
    b.release();
    if (hipMalloc(&b.p, bytes) != hipSuccess) return false;
    b.cap = bytes;
    return true;
}

// OptixDenoiserManager::denoise() replacement (csrc/hjr_denoise.hip.h), device buffers, asynchronous on `hip_stream`
extern "C" int hjr_denoise_device(hjr_ctx* c, int render_mode, uint32_t in_w, uint32_t in_h, const void* d_color, const void* d_albedo,
                                  const void* d_normal, void* d_out, uint32_t out_w, uint32_t out_h, void* hip_stream)
{
    if (!c || !d_color || !d_out) { set_error("hjr_denoise: null argument"); return HJR_ERR_ARG; }
    if (in_w == 0 || in_h == 0 || in_w > 16384 || in_h > 16384) { set_error("hjr_denoise: bad input size"); return HJR_ERR_ARG; }
    const bool up = render_mode == HJR_MODE_DENOISE_UPSCALE2X;
    if (render_mode != HJR_MODE_DEFAULT && render_mode != HJR_MODE_DENOISE && !up) { set_error("hjr_denoise: unknown render mode"); return HJR_ERR_ARG; }
    if (!up && (out_w != in_w || out_h != in_h)) { set_error("hjr_denoise: output size must equal the input size in this mode"); return HJR_ERR_ARG; }
    if (up && (out_w / 2u != in_w || out_h / 2u != in_h)) { set_error("hjr_denoise: DenoiseUpScale2X renders at (out_w / 2, out_h / 2)"); return HJR_ERR_ARG; }
    if (render_mode != HJR_MODE_DEFAULT && (!d_albedo || !d_normal)) { set_error("hjr_denoise: the albedo and normal guide AOVs are required"); return HJR_ERR_ARG; }
    HIPCHK(hipSetDevice(c->device));
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : c->stream;
    const size_t in_bytes = (size_t)in_w * in_h * 16;
    if (render_mode == HJR_MODE_DEFAULT) { // blendFactor 1: the output is the input (denoiser.h:94-97)
        if (d_out != d_color) HIPCHK(hipMemcpyAsync(d_out, d_color, in_bytes, hipMemcpyDeviceToDevice, st));
        return HJR_OK;
    }
    if (!ensure(c->d_dn_a, in_bytes) || !ensure(c->d_dn_b, in_bytes)) { set_error("hjr_denoise: allocation failed"); return HJR_ERR_DEVICE; }
    const dim3 block(256), grid((in_w + 63) / 64, (in_h + 3) / 4);
    const float4* src = (const float4*)d_color;
    float4* pp[2] = { (float4*)c->d_dn_a.p, (float4*)c->d_dn_b.p };
    for (int it = 0; it < HJR_ATROUS_PASSES; it++) {
        float4* dst = (!up && it == HJR_ATROUS_PASSES - 1) ? (float4*)d_out : pp[it & 1];
        hipLaunchKernelGGL(hjr_atrous_kernel, grid, block, 0, st, src, (const float4*)d_normal, (const float4*)d_albedo, dst, (int)in_w, (int)in_h,
                           1 << it, it < 2 ? 0.0f : 1.0f / (float)(1 << (it - 2))); // colour term: off, off, 1, 0.5, 0.25
        src = dst;
    }
    if (up) {
        const dim3 g2((out_w + 63) / 64, (out_h + 3) / 4);
        hipLaunchKernelGGL(hjr_upscale2x_kernel, g2, block, 0, st, src, (float4*)d_out, (int)in_w, (int)in_h, (int)out_w, (int)out_h);
    }
    HIPCHK(hipGetLastError());
    return HJR_OK;
}

// host-buffer form (what Renderer's frame loop does with AOV_Color / AOV_Albedo / AOV_Normal -> AOV_Output); synchronous
extern "C" int hjr_denoise(hjr_ctx* c, int render_mode, uint32_t in_w, uint32_t in_h, const float* color, const float* albedo, const float* normal,
                           float* out, uint32_t out_w, uint32_t out_h)
{
    if (!c || !color || !out) { set_error("hjr_denoise: null argument"); return HJR_ERR_ARG; }
    if (in_w == 0 || in_h == 0 || out_w == 0 || out_h == 0) { set_error("hjr_denoise: empty image"); return HJR_ERR_ARG; }
    HIPCHK(hipSetDevice(c->device));
    const size_t in_bytes = (size_t)in_w * in_h * 16, out_bytes = (size_t)out_w * out_h * 16;
    if (!ensure(c->d_color, in_bytes) || !ensure(c->d_albedo, in_bytes) || !ensure(c->d_normal, in_bytes) || !ensure(c->d_dn_out, out_bytes)) {
        set_error("hjr_denoise: allocation failed");
        return HJR_ERR_DEVICE;
    }
    HIPCHK(hipMemcpyAsync(c->d_color.p, color, in_bytes, hipMemcpyHostToDevice, c->stream));
    if (albedo) HIPCHK(hipMemcpyAsync(c->d_albedo.p, albedo, in_bytes, hipMemcpyHostToDevice, c->stream));
    if (normal) HIPCHK(hipMemcpyAsync(c->d_normal.p, normal, in_bytes, hipMemcpyHostToDevice, c->stream));
    const int rc = hjr_denoise_device(c, render_mode, in_w, in_h, c->d_color.p, albedo ? c->d_albedo.p : nullptr, normal ? c->d_normal.p : nullptr,
                                      c->d_dn_out.p, out_w, out_h, c->stream);
    if (rc != HJR_OK) return rc;
    HIPCHK(hipMemcpyAsync(out, c->d_dn_out.p, out_bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return HJR_OK;
}

// One frame of Renderer's loop in a Denoise mode, on the device: optixLaunch -> denoise -> cpyGPUBufferToHost(AOV_Output)
// (renderer.h:1229-1281).  p->width x p->height is the RENDER size (already halved by the caller for DenoiseUpScale2X).
extern "C" int hjr_render_denoised(hjr_ctx* c, const hjr_params* p_user, int render_mode, float* out, uint32_t out_w, uint32_t out_h)
{
    hjr_params params; // sized struct
    if (!c || !hjr::abi_take(p_user, params, "hjr_render_denoised")) { if (!c) set_error("hjr_render_denoised: null context"); return HJR_ERR_ARG; }
    const hjr_params* p = &params;
    if (!c || !p || !out) { set_error("hjr_render_denoised: null argument"); return HJR_ERR_ARG; }
    HIPCHK(hipSetDevice(c->device));
    const size_t in_bytes = (size_t)p->width * p->height * 16, out_bytes = (size_t)out_w * out_h * 16;
    if (in_bytes == 0 || out_bytes == 0) { set_error("hjr_render_denoised: empty image"); return HJR_ERR_ARG; }
    const bool guides = render_mode != HJR_MODE_DEFAULT;
    if (!ensure(c->d_color, in_bytes) || (guides && (!ensure(c->d_albedo, in_bytes) || !ensure(c->d_normal, in_bytes))) || !ensure(c->d_dn_out, out_bytes)) {
        set_error("hjr_render_denoised: allocation failed");
        return HJR_ERR_DEVICE;
    }
    HIPCHK(hipMemsetAsync(c->d_color.p, 0, in_bytes, c->stream));
    if (guides) { HIPCHK(hipMemsetAsync(c->d_albedo.p, 0, in_bytes, c->stream)); HIPCHK(hipMemsetAsync(c->d_normal.p, 0, in_bytes, c->stream)); }
    int rc = render_impl(c, p, c->d_color.p, guides ? c->d_albedo.p : nullptr, guides ? c->d_normal.p : nullptr, c->stream);
    if (rc != HJR_OK) return rc;
    rc = hjr_denoise_device(c, render_mode, p->width, p->height, c->d_color.p, guides ? c->d_albedo.p : nullptr, guides ? c->d_normal.p : nullptr,
                            c->d_dn_out.p, out_w, out_h, c->stream);
    if (rc != HJR_OK) return rc;
    HIPCHK(hipMemcpyAsync(out, c->d_dn_out.p, out_bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return HJR_OK;
}

// Round-trips every child ref the builder can emit for a tree that host/frame.cpp admits to the 16-bit-stack layout
// (inner node ids < HJR_STACK16_MAX_NODES; leaves of 0..HJR_STACK16_LEAF_MAX triangles starting below HJR_STACK16_MAX_TRIS)
// through stack_enc<uint16_t> / stack_dec.  Pure host code (the same inline functions the kernel uses); 0 = all refs survive.
extern "C" int hjr_selftest_stack16(void)
{
    for (uint32_t n = 0; n < HJR_STACK16_MAX_NODES; n++)
        if (stack_dec(stack_enc<uint16_t>(n)) != n) return 1;
    for (uint32_t count = 0; count <= HJR_STACK16_LEAF_MAX; count++)
        for (uint32_t first = 0; first < HJR_STACK16_MAX_TRIS; first++) {
            const uint32_t ref = HJR_LEAF_FLAG | (count << 27) | first;
            if (stack_dec(stack_enc<uint16_t>(ref)) != ref) return 2;
        }
    return 0;
}

extern "C" int hjr_pack_tiles_device(hjr_ctx* c, const void* d_frame, uint32_t w, uint32_t h, uint32_t rank, uint32_t world, void* d_packed, void* hip_stream)
{
    if (!c || !d_frame || !d_packed || w == 0 || h == 0 || world == 0 || rank >= world) { set_error("hjr_pack_tiles_device: bad argument"); return HJR_ERR_ARG; }
    HIPCHK(hipSetDevice(c->device));
    const uint32_t n = hjr_owned_tiles(w, h, rank, world);
    if (n == 0) return HJR_OK;
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : c->stream;
    hipLaunchKernelGGL(hjr_pack_tiles_kernel, dim3((unsigned)(((size_t)n * 64 + 255) / 256)), dim3(256), 0, st, (const float4*)d_frame, (float4*)d_packed, w, h, (w + HJR_TILE - 1) / HJR_TILE, n, rank, world);
    HIPCHK(hipGetLastError());
    return HJR_OK;
}
extern "C" int hjr_unpack_tiles_device(hjr_ctx* c, const void* d_packed, uint32_t w, uint32_t h, uint32_t rank, uint32_t world, void* d_frame, void* hip_stream)
{
    if (!c || !d_frame || !d_packed || w == 0 || h == 0 || world == 0 || rank >= world) { set_error("hjr_unpack_tiles_device: bad argument"); return HJR_ERR_ARG; }
    HIPCHK(hipSetDevice(c->device));
    const uint32_t n = hjr_owned_tiles(w, h, rank, world);
    if (n == 0) return HJR_OK;
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : c->stream;
    hipLaunchKernelGGL(hjr_unpack_tiles_kernel, dim3((unsigned)(((size_t)n * 64 + 255) / 256)), dim3(256), 0, st, (const float4*)d_packed, (float4*)d_frame, w, h, (w + HJR_TILE - 1) / HJR_TILE, n, rank, world);
    HIPCHK(hipGetLastError());
    return HJR_OK;
}

extern "C" int hjr_preview_device(hjr_ctx* c, const void* d_color, uint32_t w, uint32_t h, int tonemap, void* d_rgba8, void* hip_stream)
{
    if (!c || !d_color || !d_rgba8 || w == 0 || h == 0 || tonemap < HJR_TONEMAP_NONE || tonemap > HJR_TONEMAP_ACES) { set_error("hjr_preview_device: bad argument"); return HJR_ERR_ARG; }
    HIPCHK(hipSetDevice(c->device));
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : c->stream;
    const size_t n = (size_t)w * h;
    hipLaunchKernelGGL(hjr_preview_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const float4*)d_color, (uchar4*)d_rgba8, n, tonemap);
    HIPCHK(hipGetLastError());
    return HJR_OK;
}

extern "C" int hjr_synchronize(hjr_ctx* c)
{
    if (!c) { set_error("hjr_synchronize: null context"); return HJR_ERR_ARG; }
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipDeviceSynchronize());
    return HJR_OK;
}

extern "C" int hjr_get_stats(hjr_ctx* c, hjr_stats* out)
{
    if (!c || !out) { set_error("hjr_get_stats: null argument"); return HJR_ERR_ARG; }
    uint32_t out_size;
    if (!hjr::abi_size(out, out_size, "hjr_get_stats")) return HJR_ERR_ARG;
    HIPCHK(hipSetDevice(c->device));
    if (c->event_pending) {
        HIPCHK(hipEventSynchronize(c->ev1));
        float ms = 0.0f;
        HIPCHK(hipEventElapsedTime(&ms, c->ev0, c->ev1));
        c->stats.last_kernel_ms = ms;
        c->event_pending = false;
        if (c->d_work.p) { int rc = fetch_stats(c, c->stream); if (rc != HJR_OK) return rc; }
    }
    return hjr::abi_give(out, c->stats, "hjr_get_stats") ? HJR_OK : HJR_ERR_ARG; // sized struct: at most out->struct_size bytes are written
}

extern "C" int hjr_render(hjr_ctx* c, const hjr_params* p_user, float* color, float* albedo, float* normal)
{
    if (!c || !p_user || !color) { set_error("hjr_render: null argument"); return HJR_ERR_ARG; }
    hjr_params params; // sized struct
    if (!hjr::abi_take(p_user, params, "hjr_render")) return HJR_ERR_ARG;
    const hjr_params* p = &params;
    HIPCHK(hipSetDevice(c->device));
    if ((size_t)p->width * p->height == 0) { set_error("hjr_render: empty image"); return HJR_ERR_ARG; }
    // HJR_FLAG_PACKED: the buffers hold this rank's tiles only (hjr_owned_tiles x 64 float4), as in hjr_render_device
    const bool packed_out = (p->flags & HJR_FLAG_PACKED) != 0;
    if (p->rank >= (p->world_size ? p->world_size : 1u)) { set_error("hjr_render: rank >= world_size"); return HJR_ERR_ARG; }
    const size_t bytes = packed_out ? (size_t)hjr_owned_tiles(p->width, p->height, p->rank, p->world_size ? p->world_size : 1u) * 64u * 16u : (size_t)p->width * p->height * 16;
    if (bytes == 0) return HJR_OK; // a rank without tiles
    DevBuf* bufs[3] = { &c->d_color, &c->d_albedo, &c->d_normal };
    float* host[3] = { color, albedo, normal };
    for (int i = 0; i < 3; i++) {
        if (!host[i]) continue;
        if (bufs[i]->cap < bytes) {
            bufs[i]->release();
            if (hipMalloc(&bufs[i]->p, bytes) != hipSuccess) { set_error("hjr_render: AOV allocation failed"); return HJR_ERR_DEVICE; }
            bufs[i]->cap = bytes;
        }
        if (!packed_out) HIPCHK(hipMemsetAsync(bufs[i]->p, 0, bytes, c->stream));
    }
    int rc = render_impl(c, p, c->d_color.p, albedo ? c->d_albedo.p : nullptr, normal ? c->d_normal.p : nullptr, c->stream);
    if (rc != HJR_OK) return rc;
    for (int i = 0; i < 3; i++)
        if (host[i]) HIPCHK(hipMemcpyAsync(host[i], bufs[i]->p, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream)); // CUDA_SYNC_CHECK (renderer.h:1242)
    return HJR_OK;
}
