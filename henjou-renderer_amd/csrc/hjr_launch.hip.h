// Device context of the C-ABI and the launch code of the two render-kernel families.  The kernels are instantiated per integrator in
// hjr_launch_nee.hip / hjr_launch_pt.hip / hjr_launch_mis.hip (explicit instantiations of hjr_launch<I, STATS>), so that the three
// translation units compile in parallel; hjr_device.hip only declares them.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/henjou_hip.h"
#include "../host/frame.hpp"
#include "hjr_kernel.hip.h"
#include "hjr_wavefront.hip.h"

namespace hjr {
void set_error(const std::string& s);
}
using hjr::set_error;

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    bool upload(const void* src, size_t bytes, hipStream_t st)
    {
        if (bytes > cap) {
            if (p) (void)hipFree(p);
            p = nullptr; cap = 0;
            size_t want = bytes + bytes / 4 + 256;
            if (hipMalloc(&p, want) != hipSuccess) return false;
            cap = want;
        }
        if (bytes && hipMemcpyAsync(p, src, bytes, hipMemcpyHostToDevice, st) != hipSuccess) return false;
        return true;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

struct hjr_ctx {
    int device = 0;
    int n_cus = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hjr::SceneCopy scene;
    bool have_scene = false, have_frame = false;
    std::vector<float> last_m, last_inv; // instance transforms of the frame data currently on the device
    uint32_t last_build_tag = 0; // the build options the current frame data was built with
    hjr::FrameData pending; // built by hjr_prepare_transforms, made current by hjr_commit_transforms
    std::vector<float> pending_m, pending_inv;
    bool pending_valid = false, pending_same = false;
    uint32_t pending_build_tag = 0;
    double pending_build_ms = 0.0;
    hjr::FrameData frame;
    DevBuf d_nodes, d_tri_geom, d_tri_shade, d_tri_inst, d_materials, d_lights, d_lut, d_work;
    DevBuf d_texels, d_tex_desc, d_srgb_lut, d_sky;
    int sky_w = 0, sky_h = 0;
    uint32_t n_textures = 0;
    int lut_w = 0, lut_h = 0;
    DevBuf d_color, d_albedo, d_normal; // staging for hjr_render (host buffers)
    DevBuf d_part_color, d_part_albedo, d_part_normal; // chunk sums [n_chunks][H][W] float4
    DevBuf d_spill; // overflow of the short traversal stacks (memory-path kernels)
    DevBuf d_wf_ctx; // context planes of the wavefront kernel
    DevBuf d_tiles; // [tile_order | tile_class] of the cost-ordered tile list
    DevBuf d_tile_cost; // measured per-tile cost of the previous frame
    uint64_t cost_tag = 0; // (width, height, spp, rank, world, integrator) the costs belong to; 0 = none
    DevBuf d_dn_a, d_dn_b, d_dn_out; // denoise ping-pong / host-entry staging
    hjr_stats stats;
    bool event_pending = false;
    hjr::Options opt; // hjr_set_option (host/options.hpp): the library reads no environment variable
};


#ifdef HJR_FAST_MATH
#define HJR_FAST_TAG true
#else
#define HJR_FAST_TAG false
#endif
template <int I, bool S, int W> static int launch_mem(hjr_ctx* c, const KParams& kp, uint64_t n_items, hipStream_t st);
template <int I, bool S, int W, int A> static int launch_mem2(hjr_ctx* c, const KParams& kp, uint64_t n_items, hipStream_t st);
// kernel variant of a launch (VAR of hjr_render_kernel / hjr_wavefront_kernel): 2 textures / sky texture, 1 albedo / normal AOVs, 0 colour only
static int kernel_variant(const KParams& kp) { return (kp.tex_desc || kp.sky_tex) ? 2 : ((kp.aov_albedo || kp.aov_normal) ? 1 : 0); }
// lds_mode: 0 = BVH4 read from memory, 1 = BVH2 staged in LDS with 32-bit stack entries, 2 = with 16-bit entries, 3 = BVH2 from memory
template <int I, bool S, bool S16, int A> static int launch_lds2(hjr_ctx* c, const KParams& kp, uint64_t n_items, hipStream_t st)
{
    const size_t smem = (((size_t)HJR_BLOCK_LDS * kp.stack_depth * (S16 ? 2 : 4) + 15) / 16) * 16 + ((size_t)kp.n_node_f4 + kp.n_tri_f4 + kp.n_mat_f4 + kp.n_light_f4) * 16;
    auto kern = hjr_render_kernel<I, S, HJR_BLOCK_LDS, true, S16, 2, A, HJR_FAST_TAG>;
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) return -1;
    uint64_t blocks = (uint64_t)c->n_cus;
    uint64_t max_useful = (n_items + HJR_BLOCK_LDS - 1) / HJR_BLOCK_LDS;
    if (blocks > max_useful) blocks = max_useful ? max_useful : 1;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(HJR_BLOCK_LDS), smem, st, kp);
    return 0;
}
// the albedo / normal AOV sums cost 6 VGPRs per lane: a separate instantiation for callers that only want aov_color
template <int I, bool S, bool S16> static int launch_lds(hjr_ctx* c, const KParams& kp, uint64_t n_items, hipStream_t st)
{
    const int var = kernel_variant(kp);
    return var == 2 ? launch_lds2<I, S, S16, 2>(c, kp, n_items, st) : (var == 1 ? launch_lds2<I, S, S16, 1>(c, kp, n_items, st) : launch_lds2<I, S, S16, 0>(c, kp, n_items, st));
}
// Workgroup-local wavefront kernel (hjr_wavefront.hip.h): one 1024-thread workgroup per CU for every layout.  LDS holds the top of
// the traversal stacks, the scene tables (LDS layouts), the queue header, the hit slots and the id rings; what is left after the
// fixed parts decides how many stack entries per lane stay in LDS (the rest overflows to HBM).  Returns -2 when the layout does not fit.
template <int I, bool S, bool LDS, bool SP, int W, int A> static int launch_wf3(hjr_ctx* c, const KParams& kp, uint64_t n_items, uint32_t cap, uint32_t lds_entries, size_t smem, hipStream_t st)
{
    auto kern = hjr_wavefront_kernel<I, S, HJR_BLOCK_LDS, LDS, SP, W, A>;
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) return -1;
    uint64_t blocks = (uint64_t)c->n_cus;
    const uint64_t max_useful = (n_items + cap - 1) / cap;
    if (blocks > max_useful) blocks = max_useful ? max_useful : 1;
    KParams k2 = kp;
    k2.wf_cap = cap;
    k2.wf_refill = (uint32_t)c->opt.get(hjr::OPT_WF_REFILL, HJR_WF_REFILL); // tuning options
    k2.wf_trace_min = (uint32_t)c->opt.get(hjr::OPT_WF_TRACE_MIN, HJR_WF_TRACE_MIN);
    k2.wf_prefetch_min = (uint32_t)c->opt.get(hjr::OPT_WF_PREFETCH_MIN, HJR_WF_PREFETCH_MIN);
    const size_t ctx_bytes = (size_t)(HJR_WF_CTX_F4 + (A ? HJR_WF_AOV_F4 : 0)) * 16 * blocks * cap; // context records, then (albedo / normal launches) the AOV sums
    if (c->d_wf_ctx.cap < ctx_bytes) {
        c->d_wf_ctx.release();
        if (hipMalloc(&c->d_wf_ctx.p, ctx_bytes) != hipSuccess) return -1;
        c->d_wf_ctx.cap = ctx_bytes;
    }
    k2.wf_ctx = (float4*)c->d_wf_ctx.p;
    k2.wf_aov = A ? k2.wf_ctx + (size_t)HJR_WF_CTX_F4 * blocks * cap : nullptr;
    k2.stack_lds_entries = lds_entries;
    k2.spill_stride = (uint32_t)(blocks * HJR_BLOCK_LDS);
    const uint32_t over = kp.stack_depth > lds_entries ? kp.stack_depth - lds_entries : 0u;
    const size_t spill_bytes = (size_t)k2.spill_stride * (over ? over : 1u) * 4;
    if (c->d_spill.cap < spill_bytes) {
        c->d_spill.release();
        if (hipMalloc(&c->d_spill.p, spill_bytes) != hipSuccess) return -1;
        c->d_spill.cap = spill_bytes;
    }
    k2.stack_spill = (uint32_t*)c->d_spill.p;
    c->stats.stack_lds_entries = lds_entries;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(HJR_BLOCK_LDS), smem, st, k2);
    return 0;
}
// LDS holds the (top of the) traversal stacks, the scene tables (LDS layouts), the queue header and the id rings; what is left after the
// fixed parts decides how many stack entries per lane stay in LDS.  Returns -2 when the layout does not fit.
template <int I, bool S, bool LDS, int W, int A> static int launch_wf2(hjr_ctx* c, const KParams& kp, uint64_t n_items, hipStream_t st)
{
    uint32_t cap = LDS ? 2048 : 4096; // contexts per workgroup: more of them in flight pay when every node comes from memory (1 M triangles: 272 -> 259 ms)
    { const int v = c->opt.get(hjr::OPT_WF_CAP, (int)cap); if ((v & (v - 1)) == 0) cap = (uint32_t)v; }
    const bool force_short = c->opt.is_set(hjr::OPT_SHORT_STACK);
    const uint32_t short_stack = (uint32_t)c->opt.get(hjr::OPT_SHORT_STACK, HJR_SHORT_STACK);
    const size_t scene_bytes = LDS ? ((size_t)kp.n_node_f4 + kp.n_tri_f4 + kp.n_mat_f4 + kp.n_light_f4) * 16 : 0;
    const size_t fixed = scene_bytes + 96 + (size_t)HJR_WF_QUEUES * cap * 2;
    const size_t lds_max = 160u * 1024u;
    if (fixed + (size_t)HJR_BLOCK_LDS * 4 * 4 > lds_max) return -2; // not even four stack entries per lane fit
    uint32_t lds_entries = (uint32_t)((lds_max - fixed) / ((size_t)HJR_BLOCK_LDS * 4));
    if (lds_entries > kp.stack_depth) lds_entries = kp.stack_depth;
    if ((!LDS || force_short) && lds_entries > short_stack) lds_entries = short_stack;
    const size_t smem = (size_t)HJR_BLOCK_LDS * lds_entries * 4 + fixed;
    if (LDS && lds_entries >= kp.stack_depth) return launch_wf3<I, S, LDS, false, W, A>(c, kp, n_items, cap, lds_entries, smem, st); // whole stacks in LDS
    return launch_wf3<I, S, LDS, true, W, A>(c, kp, n_items, cap, lds_entries, smem, st);
}
template <int I, bool S, bool LDS, int W> static int launch_wf1(hjr_ctx* c, const KParams& kp, uint64_t n_items, hipStream_t st)
{
    const int var = kernel_variant(kp);
    return var == 2 ? launch_wf2<I, S, LDS, W, 2>(c, kp, n_items, st) : (var == 1 ? launch_wf2<I, S, LDS, W, 1>(c, kp, n_items, st) : launch_wf2<I, S, LDS, W, 0>(c, kp, n_items, st));
}
template <int I, bool S> static int launch_wf(hjr_ctx* c, const KParams& kp, uint64_t n_items, int lds_mode, hipStream_t st)
{
    if (lds_mode == 1 || lds_mode == 2) return launch_wf1<I, S, true, 2>(c, kp, n_items, st); // BVH2 + tables staged in LDS (stack entries are always 32-bit here)
    if (lds_mode == 3) return launch_wf1<I, S, false, 2>(c, kp, n_items, st);
    return launch_wf1<I, S, false, 4>(c, kp, n_items, st);
}
// descent loops of the fused traversals (hjr_traverse.hip.h): lanes still descending below which a pass moves on to the leaves
#ifndef HJR_NODE_MIN_LDS
#define HJR_NODE_MIN_LDS 6     /* megakernel, LDS-resident scenes (round 2, with AOVs, 1 / 4 / 8 / 12 / 16: 139.8 / 128.9 / 129.8 / 135.2 / 140.4 ms; round 3 with a carry-over of 14 lanes, 4 / 5 / 6 / 7 / 8: 110.1 / 109.6 / 109.15 / 109.05 / 109.3) */
#endif
#ifndef HJR_NODE_MIN_LDS_WF
#define HJR_NODE_MIN_LDS_WF 8  /* wavefront kernel, LDS-resident scenes (1 / 4 / 8 / 12 / 16: 132.4 / 125.4 / 124.8 / 125.7 / 126.4 ms) */
#endif
#ifndef HJR_NODE_MIN_MEM
#define HJR_NODE_MIN_MEM 24    /* scenes read from memory (1 M triangles, 1 / 8 / 16 / 24 / 32: megakernel 280 / 197 / 180 / 179 / 190 ms, wavefront 255 / 213 / 194 / 189 / 192) */
#endif
#ifndef HJR_TOP_NODES
#define HJR_TOP_NODES 85 /* memory layouts, BVH4: nodes of the top of the tree (levels 0 - 3) staged in LDS per workgroup (option "top_nodes"; 1 M triangles, 0 / 21 / 85 / 140 / 200 / 341: 166.9 / 165.9 / 164.6 / 164.5 / 164.6 / 273.7 ms — the last one loses a workgroup per CU) */
#endif
#ifndef HJR_HOLD_MIN
#define HJR_HOLD_MIN 8 /* megakernel: lanes of the rare material class (multiple-scattering GGX) a wave collects before it shades them (0: never hold; C2 with AOVs, 0 / 4 / 8 / 16 / 32: 129.0 / 126.5 / 126.2 / 127.9 / 144.2 ms) */
#endif
#ifndef HJR_HOLD_AGE
#define HJR_HOLD_AGE 2 /* ... or rounds the oldest of them has waited */
#endif
template <int I, bool S> int hjr_launch(hjr_ctx* c, const KParams& kp_in, uint64_t n_items, int lds_mode, hipStream_t st)
{
    KParams kp = kp_in;
    const uint32_t nm_forced = (uint32_t)c->opt.get(hjr::OPT_NODE_MIN, 0); // option "node_min"
    // Two kernel families produce the same bits (hjr_kernel.hip.h / hjr_wavefront.hip.h); which one is faster depends on the launch
    // (MI355X, profiles/r02_experiments.md §4).  Bundled scene (LDS-resident), 1080p x 256 spp: MIS 193 ms wavefront vs 234 ms megakernel
    // (the NEE shadow ray and the next closest-hit ray of its bounce are traced by sorted, full waves), NEE colour-only 126.7 vs 126.6,
    // NEE with albedo / normal AOVs 145.7 vs 128.9, Pathtrace 104.7 vs 91.2.  Scenes read from memory (1 M triangles, 1080p x 64 spp):
    // MIS 416 vs 635 ms, NEE 188 vs 179.  So: MIS -> wavefront kernel, everything else -> megakernel.  option "pipeline" overrides.
    const int pe = c->opt.get(hjr::OPT_PIPELINE, 0); // option "pipeline": 1 megakernel, 2 wavefront kernel
    const bool lds_layout = lds_mode == 1 || lds_mode == 2;
    bool wf = I == HJR_INTEGRATOR_MIS;
    if (pe == 2) wf = true;
    if (pe == 1) wf = false;
    // the wavefront kernel's queue positions are free-running 32-bit counters per workgroup (hjr_wavefront.hip.h::WfShared): a context is
    // queued at most ~12 times per sample; frames that could bring one workgroup near 2^32 pushes (4x its even share) stay with the megakernel
    if ((double)n_items * kp.chunk_spp * 12.0 * 4.0 / (double)(c->n_cus > 0 ? c->n_cus : 1) >= 4.0e9) wf = false;
    c->stats.pipeline = wf ? 1u : 0u;
    kp.hold_min = (uint32_t)c->opt.get(hjr::OPT_HOLD_MIN, HJR_HOLD_MIN); kp.hold_age = (uint32_t)c->opt.get(hjr::OPT_HOLD_AGE, HJR_HOLD_AGE); // tuning options
    kp.node_min = nm_forced ? nm_forced : (lds_layout ? (wf ? HJR_NODE_MIN_LDS_WF : HJR_NODE_MIN_LDS) : HJR_NODE_MIN_MEM);
#ifdef HJR_LEAN_VARIANT /* kernel experiments (make variant X="-DHJR_LEAN_VARIANT ..."): only the LDS-resident megakernel is instantiated: builds in seconds */
    c->stats.pipeline = 0u;
    if (!nm_forced) kp.node_min = lds_layout ? HJR_NODE_MIN_LDS : HJR_NODE_MIN_MEM;
    return lds_mode == 1 ? launch_lds<I, S, false>(c, kp, n_items, st) : (lds_mode == 0 ? launch_mem<I, S, 4>(c, kp, n_items, st) : -1);
#else
    if (wf) {
        const int rc = launch_wf<I, S>(c, kp, n_items, lds_mode, st);
        if (rc != -2) return rc;
        c->stats.pipeline = 0u; // the scene tables + queues do not fit LDS in this layout: megakernel
        if (!nm_forced) kp.node_min = lds_layout ? HJR_NODE_MIN_LDS : HJR_NODE_MIN_MEM;
    }
    if (lds_mode == 1) return launch_lds<I, S, false>(c, kp, n_items, st);
    if (lds_mode == 2) return launch_lds<I, S, true>(c, kp, n_items, st);
    if (lds_mode == 3) return launch_mem<I, S, 2>(c, kp, n_items, st); // BVH2 read from memory (option "bvh_width" = 2 on a big scene)
    return launch_mem<I, S, 4>(c, kp, n_items, st);
#endif
}
template <int I, bool S, int W> static int launch_mem(hjr_ctx* c, const KParams& kp, uint64_t n_items, hipStream_t st)
{
    const int var = kernel_variant(kp);
    return var == 2 ? launch_mem2<I, S, W, 2>(c, kp, n_items, st) : (var == 1 ? launch_mem2<I, S, W, 1>(c, kp, n_items, st) : launch_mem2<I, S, W, 0>(c, kp, n_items, st));
}
template <int I, bool S, int W, int A> static int launch_mem2(hjr_ctx* c, const KParams& kp, uint64_t n_items, hipStream_t st)
{
    const uint32_t short_stack = (uint32_t)c->opt.get(hjr::OPT_SHORT_STACK, HJR_SHORT_STACK); // tests force the overflow path with 2
    const uint32_t lds_entries = kp.stack_depth < short_stack ? kp.stack_depth : short_stack;
    // BVH4: the top of the tree (breadth-first ids: the first nodes) in LDS next to the stacks; sized so that four workgroups still share a CU
    uint32_t n_top = 0;
    if (W == 4) {
        const uint32_t want = (uint32_t)c->opt.get(hjr::OPT_TOP_NODES, HJR_TOP_NODES);
        const uint32_t have = kp.n_node_f4 / HJR_NODE4_F4;
        n_top = want < have ? want : have;
    }
    const size_t smem = (((size_t)HJR_BLOCK * lds_entries * 4 + 15) / 16) * 16 + (size_t)n_top * HJR_NODE4_F4 * 16;
    auto kern = hjr_render_kernel<I, S, HJR_BLOCK, false, false, W, A, HJR_FAST_TAG>;
    int per_cu = 0;
    if (c->opt.is_set(hjr::OPT_BLOCKS_PER_CU)) per_cu = c->opt.get(hjr::OPT_BLOCKS_PER_CU, 0);
    else if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)kern, HJR_BLOCK, smem) != hipSuccess || per_cu < 1)
        per_cu = 2;
    uint64_t blocks = (uint64_t)c->n_cus * (uint64_t)per_cu;
    uint64_t max_useful = (n_items + HJR_BLOCK - 1) / HJR_BLOCK;
    if (blocks > max_useful) blocks = max_useful ? max_useful : 1;
    KParams k2 = kp;
    k2.spill_stride = (uint32_t)(blocks * HJR_BLOCK);
    const uint32_t over = kp.stack_depth > lds_entries ? kp.stack_depth - lds_entries : 0u;
    const size_t spill_bytes = (size_t)k2.spill_stride * (over ? over : 1u) * 4;
    if (c->d_spill.cap < spill_bytes) {
        c->d_spill.release();
        if (hipMalloc(&c->d_spill.p, spill_bytes) != hipSuccess) return -1;
        c->d_spill.cap = spill_bytes;
    }
    k2.stack_spill = (uint32_t*)c->d_spill.p;
    k2.stack_lds_entries = lds_entries;
    k2.n_top_nodes = n_top;
    c->stats.stack_lds_entries = lds_entries;
    if (smem > 48 * 1024 && hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) return -1;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(HJR_BLOCK), smem, st, k2);
    return 0;
}

#ifdef HJR_FAST_MATH
// HJR_FLAG_FAST_MATH launches: the megakernel family of this (approximate-arithmetic) translation unit, every layout; no counting variant
template <int I> int hjr_launch_fast(hjr_ctx* c, const KParams& kp_in, uint64_t n_items, int lds_mode, hipStream_t st)
{
    KParams kp = kp_in;
    const bool lds_layout = lds_mode == 1 || lds_mode == 2;
    c->stats.pipeline = 0u;
    kp.hold_min = (uint32_t)c->opt.get(hjr::OPT_HOLD_MIN, HJR_HOLD_MIN); kp.hold_age = (uint32_t)c->opt.get(hjr::OPT_HOLD_AGE, HJR_HOLD_AGE);
    kp.node_min = (uint32_t)c->opt.get(hjr::OPT_NODE_MIN, lds_layout ? HJR_NODE_MIN_LDS : HJR_NODE_MIN_MEM);
    if (lds_mode == 1) return launch_lds<I, false, false>(c, kp, n_items, st);
    if (lds_mode == 2) return launch_lds<I, false, true>(c, kp, n_items, st);
    if (lds_mode == 3) return launch_mem<I, false, 2>(c, kp, n_items, st);
    return launch_mem<I, false, 4>(c, kp, n_items, st);
}
#endif
