// The Henjou hot path as one persistent-wavefront HIP megakernel for gfx950 (CDNA4):
//   ray generation -> software BVH2 traversal / triangle test -> BSDF (Disney + thin-film LUT, negative-IOR glass,
//   multiple-scattering GGX) -> next-event-estimation integrator (also Pathtrace / MIS).
//
// Execution model
//   * a work item is one pixel's run of 8 consecutive samples, executed in order by one lane, so every per-pixel fp32 sum has
//     a fixed order (bitwise independent of scheduling, tile sharding and GPU count);
//   * wavefronts are persistent: a wave takes 64 items at a time from a global queue with one atomic and hands them to its
//     idle lanes with ballot + mbcnt prefix sums; tiles are queued expensive first (cost-ordered tile list);
//   * paths are regenerated in place: a lane whose path ended (Russian roulette, miss, light hit, depth cap)
//     starts its next sample in the same loop iteration, so every trace runs with (nearly) full waves;
//   * the NEE shadow ray of a bounce is traced fused with the next closest-hit ray; slow lanes carry their traversal over
//     to the next round instead of holding the wave;
//   * the traversal stack is per lane in LDS ([level][lane] -> conflict-free ds_read/ds_write_b32); small scenes keep the
//     whole BVH2, the triangles and the material / light tables in LDS as well;
//   * nodes / triangles / shading records / materials / lights are 16-byte-record arrays fetched as dwordx4.
// Headers: hjr_params (KParams), hjr_sampling (CMJ, frames), hjr_bsdf (materials), hjr_traverse (ray traversal), this file
// (hit programs, light sampling, the megakernel, tile pre-pass, finalize).
//
// Each device function cites the reference lines it restates (paths relative to the reference's include/).
#pragma once
#include "hjr_params.hip.h"
#include "hjr_sampling.hip.h"
#include "hjr_bsdf.hip.h"
#include "hjr_traverse.hip.h"

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
        // normal map (Material.normal_tex, NonColor, bound at renderer.h:680; the code that sampled it lived in the missing closest-hit
        // program: build-defined, glTF 2.0 tangent-space semantics).  Tangent frame per triangle from the world-space edges and the uv
        // deltas, Gram-Schmidt against the normalised shading normal; n = normalize(T nx + B ny + N nz), (nx, ny, nz) = 2 texel - 1.
        // A triangle without a uv parametrisation (zero uv determinant) keeps its interpolated normal.
        const int nm_tex = (int)f2bits(m[4].y);
        if (nm_tex >= 0) {
            const float tu = s0.w * w0 + s2.w * h.b1 + s3.y * h.b2;
            const float tv = s1.w * w0 + s3.x * h.b1 + s3.z * h.b2;
            const f3 e1 = V(g0.w, g1.x, g1.y) - V(g0.x, g0.y, g0.z), e2 = V(g1.z, g1.w, g2.x) - V(g0.x, g0.y, g0.z);
            const float du1 = s2.w - s0.w, dv1 = s3.x - s1.w, du2 = s3.y - s0.w, dv2 = s3.z - s1.w;
            const float det = du1 * dv2 - du2 * dv1;
            if (det != 0.0f) {
                const float r = 1.0f / det;
                const f3 tg = (e1 * dv2 - e2 * dv1) * r, bt = (e2 * du1 - e1 * du2) * r;
                const f3 ns = normalize(prd.normal);
                const f3 tp = normalize(tg - ns * dot(ns, tg));
                f3 bp = cross(ns, tp);
                if (dot(bp, bt) < 0.0f) bp = -bp;
                const f3 e = tex_fetch(P, nm_tex, tu, tv);
                const f3 nm = V(2.0f * e.x - 1.0f, 2.0f * e.y - 1.0f, 2.0f * e.z - 1.0f);
                const f3 mapped = tp * nm.x + bp * nm.y + ns * nm.z;
                const float l2 = dot(mapped, mapped);
                if (l2 > 0.0f && l2 - l2 == 0.0f) prd.normal = mapped * (1.0f / sqrtf(l2)); // degenerate frames (NaN / zero) keep the interpolated normal
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
HD f3 light_sample(const KParams& P, const float4* lights, CMJState& st, float& pdf, float& inv_pdf, f3& normal, f3& emission)
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
    inv_pdf = l5.w; // 1.0f / pdf, divided once per frame by the host
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
    uint32_t rng_depth; // CMJState.depth
    uint32_t rng_head;  // CMJState.head: a function of (frame, spp, s, seed, pixel), computed when the sample starts.  The megakernel carries
                        // it in a register (two hash prefixes less per round); the wavefront kernel's context record has no room for it and
                        // rebuilds it when it loads a context (path_head)
    int depth;
};

// CMJState of the path that is running sample s of pixel (px, py) (build-defined seeding)
HD CMJState path_rng(const KParams& P, uint32_t px, uint32_t py, uint32_t s, uint32_t depth)
{
    return cmj_state((unsigned long long)P.frame * (unsigned long long)P.spp + (unsigned long long)s, P.seed, px + py * P.width, depth);
}
HD uint32_t path_head(const KParams& P, uint32_t px, uint32_t py, uint32_t s) { return path_rng(P, px, py, s, 0u).head; }
// the same state from a stored head: index = (frame * spp + s) % 16 needs the low bits of the sum only
HD CMJState path_rng_from_head(const KParams& P, uint32_t head, uint32_t s, uint32_t depth)
{
    CMJState st;
    st.head = head;
    st.index = (P.frame * P.spp + s) & 15u;
    st.depth = depth;
    return st;
}

// __raygen__rg for one sample (build-defined; SURVEY §8a a1/a2, stale ptx:33-106): CMJ stream keyed by
// (frame * spp + s, seed, pixel), first 2-D draw = sub-pixel jitter, u = (2(x+jx) - W) / H, v = (2(y+jy) - H) / H
HD void start_path(const KParams& P, PathState& ps, uint32_t px, uint32_t py, uint32_t s)
{
    CMJState st = path_rng(P, px, py, s, 0u);
    ps.rng_head = st.head;
    f2 j = cmj_2d(st);
    ps.rng_depth = st.depth;
    float W = (float)P.width, H = (float)P.height;
    float u = (2.0f * ((float)px + j.x) - W) / H;
    float v = (2.0f * ((float)py + j.y) - H) / H;
    f3 cd = V(P.cam_dir[0], P.cam_dir[1], P.cam_dir[2]);
    f3 cu = V(P.cam_up[0], P.cam_up[1], P.cam_up[2]);
    f3 cr = V(P.cam_right[0], P.cam_right[1], P.cam_right[2]);
    // the ray origin is the (wave-uniform) camera position: not stored per lane, see LaneCtx::fresh
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

// ------------------------------------------------------------------ per-lane path context and the two halves of a bounce
// Everything a lane carries from one bounce to the next.  The megakernel keeps it in registers for the lifetime of the lane;
// the workgroup-local wavefront kernel (hjr_wavefront.hip.h) parks it in memory between its trace and shade stages.  Both run
// the SAME two functions on it — bounce_pre_trace (Russian roulette, ray-queue refill, path regeneration) and
// bounce_post_trace (shadow-ray resolution, closest-hit / miss program, NEE / MIS, BSDF sampling) — so every pixel is the same
// bit pattern whichever kernel produced it.
struct LaneCtx {
    bool has_item, dead, path_live;
    bool fin_pending;   // a finished path whose L still waits for its last NEE shadow ray
    bool write_pending; // the item's last sample is finished; sums go out once fin_pending is resolved
    bool sh_valid;      // pending NEE shadow ray of the bounce shaded last
    bool fresh;         // the closest-hit ray of this lane starts at the camera (ps.ro is then the origin of the pending shadow ray only)
    uint32_t item;      // px | py << 13 | chunk << 26
    uint32_t s;         // current sample of the item's run
    uint32_t it_cost;   // closest-hit rays traced for the current item (feeds the measured-cost tile order)
    f3 sumL, sumA, sumN;
    float4* aov;        // wavefront kernel, albedo / normal launches: the item's (sumA, sumN) live in memory here instead of in sumA / sumN (nullptr: in registers)
    f3 sh_d, sh_contrib; // pending shadow ray: direction and the contribution that is added iff it is unoccluded
    f3 mis_contrib;      // MIS only: the BSDF-sampled light term of the bounce shaded last, added right after the pending NEE term is resolved
    float sh_tmax;
    // Register diet (the LDS variant runs at 128 VGPRs): pixel and chunk share one word; ps.ro doubles as the origin of the pending
    // shadow ray (a regenerated path starts at the wave-uniform camera position: `fresh`); ps.L keeps the finished path's radiance
    // while fin_pending (the new path's L is 0 until that is resolved).
    PathState ps;
};
HD void ctx_reset(LaneCtx& c)
{
    c.has_item = c.dead = c.path_live = c.fin_pending = c.write_pending = c.sh_valid = false;
    c.fresh = true;
    c.item = c.s = c.it_cost = 0u;
    c.aov = nullptr;
    c.sumL = c.sumA = c.sumN = c.sh_d = c.sh_contrib = c.mis_contrib = V1(0.0f);
    c.sh_tmax = 0.0f;
    c.ps.ro = c.ps.rd = c.ps.thr = c.ps.L = V1(0.0f);
    c.ps.depth = 0; c.ps.rng_depth = 0; c.ps.rng_head = 0;
}
#define HJR_PX(c) ((c).item & 0x1fffu)
#define HJR_PY(c) (((c).item >> 13) & 0x1fffu)
#define HJR_CHUNK(c) ((c).item >> 26)
#define HJR_S_END(c) min((HJR_CHUNK(c) + 1u) * P.chunk_spp, P.spp)

// Where a wave's idle lanes get their work items from.
//  * megakernel: this wave's PRIVATE range of the global queue (next / end / exhausted, all wave-uniform): 64 items per atomic on
//    the global head.  A lane stays with its wave, so whatever the wave fetched is consumed by the wave.
//  * wavefront kernel (`shared` set): contexts wander between the waves of a workgroup, so a private range could strand items with
//    a wave that no longer meets a context in need.  There the range is SHARED by the workgroup (LDS): claims are 64-bit
//    compare-and-swaps on a packed (next, end) word, the wave that finds it empty refills it from the global head under a small
//    lock, and "the global queue is dry and the shared range is empty" is then a fact of the whole workgroup.
struct SharedRange {
    unsigned long long range; // next | end << 32
    uint32_t lock;            // 1 while a wave is fetching the next range
    uint32_t exhausted;       // the global queue has run dry (set under the lock)
};
struct WaveRange {
    uint32_t next, end; bool exhausted; SharedRange* shared;
    // megakernel: a private range is 64 aligned items = ONE (tile, sample chunk).  Its decode — two integer divisions, the tile-order
    // look-up, the tile's diagonal id turned back into (x, y) — is done once when the range is fetched, wave-uniform, not by every lane
    // at every refill (a wave refills some lane in nearly every round: ~120 instructions per round): the item word of the range's
    // pixel (0, 0)
    uint32_t base_item;
};
#ifndef HJR_WF_ITEM_FETCH
#define HJR_WF_ITEM_FETCH 256u /* items per refill of a workgroup's shared range */
#endif
// Claims up to n items of the workgroup's shared range for the calling wave (ONE lane calls this): at most two runs of
// consecutive items, [a0, a0 + n0) and [a1, a1 + n1).  Returns true ("dry") when it stopped because the frame has no more items for
// this workgroup; n0 + n1 < n with false means that other waves drained the refilled range first — under contention two runs can both
// be short — and the lanes left without an item must ask again (they are NOT finished: the wavefront kernel re-queues them).
HD bool shared_claim(const KParams& P, SharedRange* S, uint32_t n, uint32_t& a0, uint32_t& n0, uint32_t& a1, uint32_t& n1)
{
    a0 = a1 = n0 = n1 = 0u;
    uint32_t left = n;
    bool dry = false;
    while (left) {
        const unsigned long long old = __hip_atomic_load(&S->range, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const uint32_t nx = (uint32_t)old, en = (uint32_t)(old >> 32);
        if (en != nx) {
            const uint32_t take = min(left, en - nx);
            if (atomicCAS(&S->range, old, (unsigned long long)(nx + take) | ((unsigned long long)en << 32)) != old) continue;
            if (n0 == 0u) { a0 = nx; n0 = take; } else { a1 = nx; n1 = take; }
            left -= take;
            if (n1) break; // two runs are all a caller can take; whoever is still without an item asks again
            continue;
        }
        if (__hip_atomic_load(&S->exhausted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) { dry = true; break; }
        if (atomicCAS(&S->lock, 0u, 1u) == 0u) { // this wave refills (unless another one just did)
            const unsigned long long now = __hip_atomic_load(&S->range, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if ((uint32_t)now == (uint32_t)(now >> 32)) {
                const uint32_t base = atomicAdd(P.queue_head, HJR_WF_ITEM_FETCH);
                if (base >= P.n_owned_items) __hip_atomic_store(&S->exhausted, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                else __hip_atomic_store(&S->range, (unsigned long long)base | ((unsigned long long)min(base + HJR_WF_ITEM_FETCH, P.n_owned_items) << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            __hip_atomic_store(&S->lock, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else __builtin_amdgcn_s_sleep(2);
    }
    return dry;
}

// Launch parameters that only the start and the end of a work item need (tile list, queue head, output and chunk-sum pointers, shard
// geometry) are read from the kernel-argument segment WHERE they are used instead of living in SGPRs for the whole kernel: the render
// loop keeps ~100 scalar values alive, and every one of these that stays resident pushes another into a VGPR lane (v_writelane at entry,
// v_readlane at each use: VALU instructions inside the round loop).  The empty asm makes the base pointer opaque, so the compiler can
// neither hoist the load out of the loop nor merge it with the by-value copy of P.
template <typename T> HD T cold_param(size_t offset)
{
    const char __attribute__((address_space(4)))* ka = (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ka));
    return *reinterpret_cast<const T __attribute__((address_space(4)))*>(ka + offset);
}
#define HJR_COLD(field) cold_param<decltype(KParams::field)>(offsetof(KParams, field))

// NaN/Inf guard + ordered accumulation of one finished sample
template <bool STATS> HD void finish_sample(const KParams& P, LaneCtx& c, f3 L, unsigned long long* lc, const uint32_t s_back = 0u) // s_back = 1: the sample before c.s (its bookkeeping was closed while the shadow ray was pending)
{
    float sum = L.x + L.y + L.z;
    if (!(sum - sum == 0.0f)) {
        L = V1(0.0f);
        if (STATS) { // counting launches also note WHERE (the first HJR_NAN_LIST of them, in no particular order): x | y << 13 | sample << 26
            lc[9] += 1;
            const unsigned long long slot = atomicAdd(&P.nan_list[0], 1ull);
            if (slot < HJR_NAN_LIST) P.nan_list[1 + slot] = (unsigned long long)HJR_PX(c) | ((unsigned long long)HJR_PY(c) << 13) | ((unsigned long long)(c.s - s_back) << 26);
        }
    }
    c.sumL = c.sumL + L;
    if (STATS) lc[0] += 1;
}
// sample bookkeeping at the end of a path (independent of the radiance value)
HD void close_sample(const KParams& P, LaneCtx& c)
{
    c.s++;
    c.path_live = false;
    if (c.s == HJR_S_END(c)) { c.write_pending = true; c.has_item = false; }
}
template <bool AOVS> HD void write_out(const KParams& P, LaneCtx& c)
{
    const float inv_spp = 1.0f / (float)P.spp;
    const uint32_t tiles_x = HJR_COLD(tiles_x), world = HJR_COLD(world), n_chunks = HJR_COLD(n_chunks); // (cold parameters: see cold_param)
    // AOV element of the pixel: row-major frame, or (HJR_FLAG_PACKED) this rank's tiles back to back: (owned tile index) * 64 + pixel in tile
    const size_t pix = HJR_COLD(packed) ? (size_t)(hjr_tile_id(HJR_PX(c) / HJR_TILE, HJR_PY(c) / HJR_TILE, tiles_x) / world) * 64u + ((HJR_PY(c) & 7u) * 8u + (HJR_PX(c) & 7u))
                                        : (size_t)HJR_PX(c) + (size_t)HJR_PY(c) * P.width;
    f3 sumA = c.sumA, sumN = c.sumN;
    if (AOVS && c.aov) { const float4 a = c.aov[0], n = c.aov[1]; sumA = V(a.x, a.y, a.z); sumN = V(n.x, n.y, n.z); }
    if (n_chunks == 1u) { // the item is the whole pixel: mean = chunk sum * (1 / spp)
        float4* const aov_color = HJR_COLD(aov_color);
        aov_color[pix] = make_float4(c.sumL.x * inv_spp, c.sumL.y * inv_spp, c.sumL.z * inv_spp, 1.0f);
        if (AOVS) {
            float4* const aov_albedo = HJR_COLD(aov_albedo);
            float4* const aov_normal = HJR_COLD(aov_normal);
            if (aov_albedo) aov_albedo[pix] = make_float4(sumA.x * inv_spp, sumA.y * inv_spp, sumA.z * inv_spp, 1.0f);
            if (aov_normal) aov_normal[pix] = make_float4(sumN.x * inv_spp, sumN.y * inv_spp, sumN.z * inv_spp, 1.0f);
        }
    } else { // chunk sum -> HBM; hjr_finalize_kernel adds the chunks of a pixel in chunk order.  The buffers hold this rank's
             // tiles only: slot = ((chunk * owned tiles) + owned tile index) * 64 + pixel in tile
        const uint32_t otile = hjr_tile_id(HJR_PX(c) / HJR_TILE, HJR_PY(c) / HJR_TILE, tiles_x) / world;
        const size_t slot = ((size_t)HJR_CHUNK(c) * HJR_COLD(n_owned_tiles) + otile) * 64u + ((HJR_PY(c) & 7u) * 8u + (HJR_PX(c) & 7u));
        float4* const part_color = HJR_COLD(part_color);
        part_color[slot] = make_float4(c.sumL.x, c.sumL.y, c.sumL.z, 0.0f);
        if (AOVS) {
            float4* const part_albedo = HJR_COLD(part_albedo);
            float4* const part_normal = HJR_COLD(part_normal);
            if (part_albedo) part_albedo[slot] = make_float4(sumA.x, sumA.y, sumA.z, 0.0f);
            if (part_normal) part_normal[slot] = make_float4(sumN.x, sumN.y, sumN.z, 0.0f);
        }
    }
    c.write_pending = false;
}

// decode of the 64 aligned items [base, base + 64) of the megakernel's queue: item q = ((owned tile * n_chunks) + chunk) * 64 + pixel-in-tile
HD uint32_t decode_range(uint32_t base, uint32_t n_chunks, uint32_t tiles_x, uint32_t world, uint32_t rank, const uint32_t* tile_order)
{
    const uint32_t tc = base >> 6;
    const uint32_t owned = tc / n_chunks, chunk = tc - owned * n_chunks;
    const uint32_t tile = tile_order ? tile_order[owned] : owned * world + rank;
    uint32_t tx, ty;
    hjr_tile_xy(tile, tiles_x, &tx, &ty);
    return (tx * HJR_TILE) | ((ty * HJR_TILE) << 13) | (chunk << 26);
}
// ---- first half of a bounce.  ALL 64 lanes of the wave must call it together (ballots and shuffles inside); `active` is
// false for lanes that sit this round out (megakernel: lanes whose traversal is carried over; wavefront: lanes without a
// context).  On return `tracing` says whether the lane has a closest-hit ray to trace (origin: camera if c.fresh, else c.ps.ro;
// direction c.ps.rd) and c.sh_valid whether a shadow ray (origin c.ps.ro, direction c.sh_d, tmax c.sh_tmax) goes with it.
template <bool STATS, bool AOVS, bool ONEWRITE = false>
HD void bounce_pre_trace(const KParams& P, LaneCtx& c, WaveRange& wr, const bool active, bool& tracing, unsigned long long* lc)
{
    const uint32_t lane = threadIdx.x & 63u;
    // ---- Russian roulette (rt.h:173-179) with in-place path regeneration: a lane whose path dies here starts its
    //      next sample immediately, so it still has a closest-hit ray for this iteration's trace.  The dead path's
    //      radiance is final only after its pending shadow ray (fused into the same trace) is resolved.
    if (active) {
        if (c.has_item && c.path_live) { // continuing path
            const float russian_p = fmaxf(c.ps.thr.x, fmaxf(c.ps.thr.y, c.ps.thr.z));
            CMJState rr = path_rng_from_head(P, c.ps.rng_head, c.s, c.ps.rng_depth);
            const float xi_rr = cmj_1d(rr);
            c.ps.rng_depth = rr.depth;
            if (russian_p < xi_rr) {
                if (c.sh_valid) { c.fin_pending = true; close_sample(P, c); } // radiance final once the pending shadow ray is resolved
                else { // nothing pending: the sample is final now, and if it was the item's last one the lane refills below
                    finish_sample<STATS>(P, c, c.ps.L, lc);
                    c.ps.L = V1(0.0f);
                    close_sample(P, c);
                    if (!ONEWRITE && c.write_pending) write_out<AOVS>(P, c);
                }
            } else c.ps.thr = c.ps.thr / russian_p;
        }
    }

    // ---- ONEWRITE (wavefront kernel): an item whose last sample is final goes out HERE and nowhere else — close_sample() only raises
    //      write_pending wherever a path ends (roulette above; miss / light hit / depth limit / resolved last shadow ray in
    //      bounce_post_trace), so the write-out code (two integer divisions, up to six stores) exists once instead of four times: MIS
    //      192.9 -> 188.2 ms.  Nothing is added to an item's sums after its last sample closed, and the refill below waits for the write.
    //      The megakernel keeps the four in-place writes: there this extra test in every round cost 0.4 - 1 %.
    if (ONEWRITE && active && c.write_pending && !c.fin_pending) write_out<AOVS>(P, c);

    // ---- ray-queue refill (ballot + mbcnt prefix): idle lanes take consecutive items from the wave's private range
    //      [wr.next, wr.end); when it runs dry the wave fetches the next 64 items with ONE atomic on the global head.
    {
        const bool need = active && !c.has_item && !c.dead && !c.write_pending && !c.fin_pending && !c.sh_valid;
        const unsigned long long m = __ballot(need);
        if (m) {
            const uint32_t n = (uint32_t)__popcll(m);
            const uint32_t prefix = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            // (cold parameters, read here and not kept in SGPRs across the render loop: see cold_param)
            const uint32_t n_owned_items = HJR_COLD(n_owned_items), n_chunks = HJR_COLD(n_chunks), tiles_x = HJR_COLD(tiles_x), world = HJR_COLD(world);
            const uint32_t* const tile_order = HJR_COLD(tile_order);
            uint32_t* const tile_cost = HJR_COLD(tile_cost);
            uint32_t q;
            uint32_t r_item = 0u; // item word of pixel (0, 0) of the range this lane's item comes from
            bool again = false; // shared range only: unserved lanes ask again in their next pass instead of retiring
            if (wr.shared) { // wavefront kernel: the workgroup's shared range
                uint32_t a0 = 0, n0 = 0, a1 = 0, n1 = 0, dry = 1u;
                if (lane == 0) dry = shared_claim(P, wr.shared, n, a0, n0, a1, n1) ? 1u : 0u;
                again = __builtin_amdgcn_readfirstlane((int)dry) == 0;
                a0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)a0); n0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)n0);
                a1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)a1); n1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)n1);
                q = prefix < n0 ? a0 + prefix : (prefix - n0 < n1 ? a1 + (prefix - n0) : 0xffffffffu);
            } else {
            q = wr.next + prefix;                   // wave-uniform wr.next / wr.end
            const uint32_t have = wr.end - wr.next; // items left in the private range
            r_item = wr.base_item;
            if (n > have) {                         // not enough: lanes beyond `have` come from a fresh range
                // once a wave has seen the queue run dry it never touches the head again (wave-uniform flag): the 32-bit head
                // overshoots n_owned_items by at most 64 per wave of the grid and cannot wrap (hjr_device.hip keeps that margin)
                uint32_t base = 0xffffffffu;
                if (!wr.exhausted) {
                    if (lane == 0) base = atomicAdd(HJR_COLD(queue_head), 64u);
                    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                    wr.exhausted = base >= n_owned_items;
                }
                if (wr.exhausted) {
                    if (prefix >= have) q = 0xffffffffu;
                    wr.next = wr.end = 0u;
                } else {
                    const uint32_t bi = (uint32_t)__builtin_amdgcn_readfirstlane((int)decode_range(base, n_chunks, tiles_x, world, HJR_COLD(rank), tile_order)); // (wave-uniform)
                    if (prefix >= have) { q = base + (prefix - have); r_item = bi; }
                    wr.next = base + (n - have);
                    wr.end = base + 64u;
                    wr.base_item = bi;
                }
            } else wr.next += n;
            }
            // measured cost of a tile (orders the tiles of the next frame, hjr_cost_hist_kernel): a lane sums the rays of its
            // consecutive items of one tile and flushes when it moves on; lanes leaving the same tile together (the usual
            // case) share one atomic.  All lanes are here (m is wave-uniform), so the shuffles below are well defined.
            if (wr.shared && need && q < n_owned_items) r_item = decode_range(q & ~63u, n_chunks, tiles_x, world, HJR_COLD(rank), tile_order); // (wavefront kernel: runs of the shared range, decoded per lane)
            if (tile_cost) {
                const uint32_t old_tile = hjr_tile_id(HJR_PX(c) / HJR_TILE, HJR_PY(c) / HJR_TILE, tiles_x);
                const uint32_t new_tile = (need && q < n_owned_items) ? hjr_tile_id((r_item & 0x1fffu) / HJR_TILE, ((r_item >> 13) & 0x1fffu) / HJR_TILE, tiles_x) : 0xffffffffu;
                bool flush = need && c.it_cost != 0u && new_tile != old_tile;
                while (__ballot(flush)) {
                    const int leader = __ffsll((long long)__ballot(flush)) - 1;
                    const uint32_t t = (uint32_t)__shfl((int)old_tile, leader);
                    const bool mine = flush && old_tile == t;
                    uint32_t v = mine ? c.it_cost : 0u;
                    for (int off = 32; off > 0; off >>= 1) v += (uint32_t)__shfl_xor((int)v, off);
                    if ((int)lane == leader) atomicAdd(&tile_cost[t / world], v);
                    if (mine) { c.it_cost = 0u; flush = false; }
                }
            }
            if (need) {
                if (q < n_owned_items) {
                    // the 64 lanes of a wave start on one tile and one sample chunk (coherent primary rays): the range's item word + the pixel in the tile
                    const uint32_t item = r_item | (q & 7u) | (((q >> 3) & 7u) << 13);
                    const uint32_t px = item & 0x1fffu, py = (item >> 13) & 0x1fffu, chunk = item >> 26;
                    if (px < P.width && py < P.height) {
                        c.has_item = true; c.path_live = false;
                        c.item = item;
                        c.s = chunk * P.chunk_spp;
                        c.sumL = V1(0.0f);
                        if (AOVS && c.aov) c.aov[0] = c.aov[1] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                        else { c.sumA = V1(0.0f); c.sumN = V1(0.0f); }
                    }
                } else if (!again) c.dead = true;
            }
        }
    }

    if (active) {
        if (c.has_item && !c.path_live) {
            start_path(P, c.ps, HJR_PX(c), HJR_PY(c), c.s);
            if (!c.fin_pending) c.ps.L = V1(0.0f); // while fin_pending, ps.L still belongs to the finished path
            c.path_live = true; c.fresh = true;
            // the new path's own roulette draw: throughput is (1,1,1), so russian_p = 1 > xi for every xi in [0,1) and
            // thr / 1 == thr; only the stream position moves
            c.ps.rng_depth += 1u;
        }
        tracing = c.has_item;
    }
}

// ---- second half of a bounce, for a lane whose rays of this round are resolved: `occluded` answers the pending shadow ray
// (if c.sh_valid), `h` the closest-hit ray (if tracing).  Lane-private: no cross-lane operation inside.
template <int INTEGRATOR, bool STATS, bool AOVS, bool TEX, int WIDTH, int BLOCK, typename ST, bool ONEWRITE = false>
HD void bounce_post_trace(const KParams& P, const float4* nodes, const float4* tris, const float4* mats, const float4* lights, LaneCtx& c,
                          const bool tracing, const bool occluded, const Hit& h, ST& stack, unsigned long long* lc)
{
    PathState& ps = c.ps;
    if (c.sh_valid) { // `if (!light_shot.is_hit) LTE += ...` (rt.h:245-259), added to the path the shadow ray belongs to
        if (!occluded) ps.L = ps.L + c.sh_contrib; // ps.L is the finished path's radiance while fin_pending
        c.sh_valid = false;
        if (INTEGRATOR == HJR_INTEGRATOR_MIS_) { ps.L = ps.L + c.mis_contrib; c.mis_contrib = V1(0.0f); } // rt.h:378 first, then :414 / :418
    }
    if (c.fin_pending) {
        finish_sample<STATS>(P, c, ps.L, lc, 1u);
        ps.L = V1(0.0f); // from here on ps.L belongs to the path that was regenerated (or to nothing)
        c.fin_pending = false;
        if (!ONEWRITE && c.write_pending) write_out<AOVS>(P, c);
    }
    if (!tracing) return;

    c.it_cost++;
    float4 aov0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), aov1 = aov0;
    if (AOVS && c.aov && ps.depth == 0) { aov0 = c.aov[0]; aov1 = c.aov[1]; } // sums in memory: the loads fly while the hit program runs
    HitInfo prd;
    hit_program<STATS, TEX>(P, tris, mats, h, ps.rd, prd, lc);
    if (AOVS && ps.depth == 0) { // rt.h:191-194
        if (c.aov) {
            c.aov[0] = make_float4(aov0.x + prd.surf.basecolor.x, aov0.y + prd.surf.basecolor.y, aov0.z + prd.surf.basecolor.z, 0.0f);
            c.aov[1] = make_float4(aov1.x + prd.normal.x, aov1.y + prd.normal.y, aov1.z + prd.normal.z, 0.0f);
        } else { c.sumA = c.sumA + prd.surf.basecolor; c.sumN = c.sumN + prd.normal; }
    }
    if (!prd.is_hit || prd.is_light) {
        // NEE / MIS count emission only at depth 0 (rt.h:196-208, 318-330); Pathtrace always (rt.h:118-126)
        if (INTEGRATOR == HJR_INTEGRATOR_PT_ || ps.depth == 0) ps.L = ps.L + ps.thr * prd.emission;
        finish_sample<STATS>(P, c, ps.L, lc);
        close_sample(P, c);
        if (!ONEWRITE && c.write_pending) write_out<AOVS>(P, c);
        return;
    }
    CMJState st = path_rng_from_head(P, ps.rng_head, c.s, ps.rng_depth);
    const Surface& sf = prd.surf;
    f3 t, b;
    const f3 n = prd.normal;
    orthonormal_basis(n, t, b);
    const f3 local_wo = world_to_local(-ps.rd, t, n, b);
    const float lambda_wo = disney_lambda_wo(sf, local_wo); // shared by every Disney evaluate / pdf of this hit

    if (INTEGRATOR != HJR_INTEGRATOR_PT_ && P.n_lights >= 1u) { // light_prim_count < 1: no contribution (UB in the reference)
        float light_pdf, inv_light_pdf;
        f3 light_color, light_normal;
        const f3 light_position = light_sample(P, lights, st, light_pdf, inv_light_pdf, light_normal, light_color);
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
        // traced fused with the next closest-hit ray in the next round
        const float cosine1 = absdot(n, sd);
        const float cosine2 = absdot(light_normal, -sd);
        const f3 local_wi = world_to_local(sd, t, n, b);
        const f3 bsdf = bsdf_eval(P, sf, local_wo, local_wi, lambda_wo);
        const float G = cosine2 / (light_distance * light_distance);
        if (INTEGRATOR == HJR_INTEGRATOR_NEE_) {
            c.sh_contrib = (ps.thr * ((bsdf * G * cosine1) * inv_light_pdf)) * light_color; // rt.h:258: float3 / light_pdf
        } else {
            const float pt_pdf = bsdf_pdf(sf, local_wo, local_wi, lambda_wo) * G;
            const float mis_weight = light_pdf / (light_pdf + pt_pdf);
            c.sh_contrib = ((ps.thr * ((bsdf * G * cosine1) * inv_light_pdf)) * mis_weight) * light_color; // rt.h:378
        }
        c.sh_d = sd; c.sh_tmax = light_distance - 0.001f; // origin = prd.position = ps.ro below
        // an exactly-zero contribution (every hit on the glass lobe, whose evaluateBSDF is 0) cannot change L whatever
        // the shadow ray returns (x + 0 == x): skip the trace.  A NaN contribution still goes through.
        c.sh_valid = !(c.sh_contrib.x == 0.0f && c.sh_contrib.y == 0.0f && c.sh_contrib.z == 0.0f);
    }

    if (INTEGRATOR == HJR_INTEGRATOR_MIS_) { // BSDF-sampled light hit, rt.h:383-420
        // MIS adds this term AFTER the NEE term of the same bounce (rt.h:378 then :414/:418).  The NEE term still waits for its shadow
        // ray, which is traced with the next closest-hit ray like NEE's (one stand-alone traversal per bounce instead of two): the term
        // computed here is parked in c.mis_contrib and added right behind the NEE term when that is resolved, so the order of the
        // float additions is the reference's.  Without a pending NEE term (exactly-zero contribution, no lights) it is added at once.
        float pt_pdf = 1.0f; // uninitialised in the reference when msGGX returns early; defined as 1
        f3 local_wi = V(0.0f, 1.0f, 0.0f);
        const f3 brdf = bsdf_sample(P, sf, local_wo, local_wi, pt_pdf, st, lambda_wo);
        const f3 wi = local_to_world(local_wi, t, n, b);
        const float cosine1 = absdot(wi, n);
        HitInfo lh;
        ray_trace<STATS, TEX, WIDTH, BLOCK, ST>(P, nodes, tris, mats, prd.position, wi, lh, stack, lc);
        bool have_term = false;
        f3 term = V1(0.0f);
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
                            const f3 cr = cross(V(a1.x, a1.y, a1.z) - V(a0.x, a0.y, a0.z), V(a2.x, a2.y, a2.z) - V(a0.x, a0.y, a0.z));
                            const float area = length3(cr) * 0.5f;
                            lp = 1.0f / (area * P.n_lights);
                            break;
                        }
                    }
                    lp = lp * invG;
                }
                const float mis_weight = pt_pdf / (pt_pdf + lp);
                term = ((((ps.thr * mis_weight) * cosine1) * lh.emission) * brdf) / pt_pdf; // rt.h:414
                have_term = true;
            }
        } else {
            term = (((ps.thr * brdf) * cosine1) * lh.emission) / pt_pdf; // rt.h:418
            have_term = true;
        }
        if (have_term) {
            if (c.sh_valid) c.mis_contrib = term;
            else ps.L = ps.L + term;
        }
    }

    float pdf = 1.0f;
    f3 local_wi = V(0.0f, 1.0f, 0.0f);
    if (INTEGRATOR != HJR_INTEGRATOR_PT_) (void)cmj_2d(st); // drawn and discarded by the reference (rt.h:266, 426)
    const f3 bsdf = bsdf_sample(P, sf, local_wo, local_wi, pdf, st, lambda_wo);
    const f3 wi = local_to_world(local_wi, t, n, b);
    ps.thr = ps.thr * ((bsdf * fabsf(dot(wi, n))) / pdf); // rt.h:274
    ps.ro = prd.position;
    ps.rd = wi;
    c.fresh = false;
    ps.rng_depth = st.depth;
    ps.depth++;
    if (ps.depth == 10) { // MaxDepth (rt.h:166): the path is over; its last shadow ray, if any, is still pending
        if (c.sh_valid) { c.fin_pending = true; close_sample(P, c); }
        else {
            finish_sample<STATS>(P, c, ps.L, lc);
            ps.L = V1(0.0f);
            close_sample(P, c);
            if (!ONEWRITE && c.write_pending) write_out<AOVS>(P, c);
        }
    }
}

// stages the scene tables of an LDS-resident layout behind the traversal stacks (all threads of the workgroup; ends with a barrier)
template <typename SE, int BLOCK>
HD void stage_scene_in_lds(const KParams& P, float4* base, const float4*& nodes, const float4*& tris, const float4*& mats, const float4*& lights)
{
    float4* l_nodes = base;
    float4* l_tris = l_nodes + P.n_node_f4;
    for (uint32_t i = threadIdx.x; i < P.n_node_f4; i += BLOCK) l_nodes[i] = P.nodes[i];
    for (uint32_t i = threadIdx.x; i < P.n_tri_f4; i += BLOCK) l_tris[i] = P.tri_geom[i];
    // the (small) material and light tables ride along: one LDS read instead of an L2 round trip per shaded hit
    float4* l_mats = l_tris + P.n_tri_f4;
    float4* l_lights = l_mats + P.n_mat_f4;
    for (uint32_t i = threadIdx.x; i < P.n_mat_f4; i += BLOCK) l_mats[i] = P.materials[i];
    for (uint32_t i = threadIdx.x; i < P.n_light_f4; i += BLOCK) l_lights[i] = P.lights[i];
    __syncthreads();
    nodes = l_nodes; tris = l_tris; mats = l_mats; lights = l_lights;
}

// ------------------------------------------------------------------ the megakernel
// VAR: 0 = lean (colour only, untextured scene, constant sky), 1 = + albedo / normal AOV sums, 2 = + material textures, normal maps
// and the equirect sky texture.  Each step costs registers (NEE, LDS layout: 20 / 25 / 48 VGPR spills), so a launch gets the
// smallest variant that does what it needs.
// FAST only tags the symbol: the HJR_FAST_MATH translation units (hjr_launch_fast_*.hip; approximate division / square root / sine / cosine /
// power in the SHADING code, traversal and triangle test unchanged) instantiate FAST = true, the exact ones FAST = false.
template <int INTEGRATOR, bool STATS, int BLOCK, bool LDSBVH, bool STACK16, int WIDTH, int VAR, bool FAST = false>
__global__ void __launch_bounds__(BLOCK, (LDSBVH ? 1 : HJR_MIN_WAVES)) hjr_render_kernel(const KParams P)
{
    constexpr bool AOVS = VAR >= 1, TEX = VAR == 2;
    typedef typename std::conditional<STACK16, uint16_t, uint32_t>::type SE; // stack entry type
    typedef LaneStack<SE, BLOCK, !LDSBVH, STATS, LDSBVH> ST;
    ST stack;
    stack.n_over = 0;
    stack.lds = reinterpret_cast<SE*>(hjr_smem) + threadIdx.x;
    stack.spill = P.stack_spill + (blockIdx.x * BLOCK + threadIdx.x);
    stack.spill_stride = P.spill_stride;
    stack.lds_n = (int)P.stack_lds_entries;
    stack.top = nullptr; stack.n_top = 0u;
    const uint32_t lane = threadIdx.x & 63u;
    const float4* nodes = P.nodes;
    const float4* tris = P.tri_geom;
    const float4* mats = P.materials;
    const float4* lights = P.lights;
    if (LDSBVH) stage_scene_in_lds<SE, BLOCK>(P, hjr_smem + (BLOCK * P.stack_depth * (uint32_t)sizeof(SE) + 15u) / 16u, nodes, tris, mats, lights);
    else if (WIDTH == 4 && P.n_top_nodes) { // memory layout: the top of the (breadth-first) BVH4 behind the short stacks
        float4* top = hjr_smem + (BLOCK * P.stack_lds_entries * (uint32_t)sizeof(SE) + 15u) / 16u;
        for (uint32_t i = threadIdx.x; i < P.n_top_nodes * HJR_NODE4_F4; i += BLOCK) top[i] = P.nodes[i];
        __syncthreads();
        stack.top = top; stack.n_top = P.n_top_nodes;
    }

    unsigned long long lc[HJR_NSTAT];
    if (STATS) for (int i = 0; i < HJR_NSTAT; i++) lc[i] = 0;

    LaneCtx c;
    ctx_reset(c);
    WaveRange wr; wr.next = wr.end = 0u; wr.exhausted = false; wr.shared = nullptr; wr.base_item = 0u;
    bool inflight = false;      // carry-over: this lane's traversal continues in the next round (it skips everything else)
    bool tracing = false, occluded = false;
    Hit h;
    TravCarry tc; tc.cur = HJR_TRAV_DONE; tc.sp = 0; tc.phase = 2;
#ifdef HJR_TIMING
    // diagnostic build: wave-clock shares of the loop's phases, summed per wave into P.stats[HJR_NSTAT..] (never in the shipped build)
    unsigned long long tk[3] = { 0, 0, 0 }, oc[4] = { 0, 0, 0, 0 }; // rounds, lanes tracing closest, lanes with a shadow ray, lanes serviced
    unsigned long long td[10] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 }; // traversal lane occupancy (hjr_traverse.hip.h): per-lane sums, added up at the end
    unsigned long long tstamp = __builtin_amdgcn_s_memtime();
#define HJR_TICK(i) { __builtin_amdgcn_sched_barrier(0); unsigned long long now_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_sched_barrier(0); tk[i] += now_ - tstamp; tstamp = now_; }
#else
#define HJR_TICK(i)
#endif

    constexpr bool HOLD = LDSBVH; // (LDS-resident scenes only: on the 1 M-triangle scene the two extra live registers cost 3 % and the hold gains nothing)
    uint32_t hold = 0u; // > 0: this lane holds a resolved hit of a rare material class back (rounds held so far), see below
    for (;;) {
        bounce_pre_trace<STATS, AOVS>(P, c, wr, !inflight && (!HOLD || hold == 0u), tracing, lc);
        if (__ballot(!c.dead) == 0ull) break; // a lane only dies with nothing pending
        HJR_TICK(0)
#ifdef HJR_TIMING
        oc[0] += 1; oc[1] += __popcll(__ballot(tracing)); oc[2] += __popcll(__ballot(c.sh_valid));
#endif
        // ---- one fused traversal: pending shadow ray (TraceOcculution, rt.h:236-243) then closest-hit ray (RayTrace, rt.h:182-189)
        {
            Counters ca, cb; ca.box = ca.tri = cb.box = cb.tri = 0;
            const f3 cam_o = V(P.cam_pos[0], P.cam_pos[1], P.cam_pos[2]);
            inflight = traverse_fused<STATS, WIDTH, BLOCK, ST, (LDSBVH ? HJR_CARRY_LDS : HJR_CARRY_MEM)>(nodes, tris, c.sh_valid, c.ps.ro, c.sh_d, c.sh_tmax, tracing, c.fresh ? cam_o : c.ps.ro, c.ps.rd, occluded, h, stack, ca, cb, inflight || (HOLD && hold != 0u), tc, P.node_min
#ifdef HJR_TIMING
                                                                                                                   , td
#endif
            );
            if (STATS) { // tests are counted round by round, rays when they are resolved
                lc[5] += ca.box; lc[6] += ca.tri; lc[3] += cb.box; lc[4] += cb.tri;
                if (!inflight && (!HOLD || hold == 0u)) { if (c.sh_valid) lc[2] += 1; if (tracing) lc[1] += 1; }
            }
        }
        HJR_TICK(1)
#ifdef HJR_TIMING
        oc[3] += __popcll(__ballot(!inflight && (tracing || c.sh_valid)));
#endif
        // Rare material class held back (P.hold_min > 0): a wave runs the multiple-scattering GGX walk (a loop of up to six ~700-instruction
        // steps) whenever ONE of its lanes has hit such a surface — about four lanes of a round on the bundled scene.  A lane with such a hit
        // waits (its resolved rays stay where the carry-over keeps them: `h`, `occluded`, flags) until the wave holds P.hold_min of them, or it
        // has waited P.hold_age rounds, or nothing else is left to shade; then they all walk together.  What a lane computes, and the order
        // of its samples, do not change.  The class comes from the untextured material: a scheduling hint only.  (Holding glass hits back
        // the same way was measured too: 127.4 vs 126.4 ms, its sampling code is short.)
        bool shade = !inflight;
        if (HOLD && P.hold_min) {
            bool rare = false;
            if (shade && tracing && h.prim != 0xffffffffu) {
                const float4* m = mats + f2bits(tris[h.k * HJR_TRI_F4 + 2].z) * HJR_MAT_F4;
                const float4 m0 = m[0], m3 = m[3];
                rare = f2bits(m3.x) == 0 && f2bits(m3.y) == 0 && m0.w > 0.5f; // not a light, not glass, metallic
            }
            const uint32_t n_rare = (uint32_t)__popcll(__ballot(rare));
            if (n_rare) {
                const bool run = n_rare >= P.hold_min || __ballot(rare && hold >= P.hold_age) != 0ull || __ballot(shade && !rare) == 0ull;
                if (rare && !run) { hold++; shade = false; tc.phase = 2; tc.cur = HJR_TRAV_DONE; tc.sp = 0; }
            }
        }
        if (shade) {
            if (HOLD) hold = 0u;
            bounce_post_trace<INTEGRATOR, STATS, AOVS, TEX, WIDTH, BLOCK, ST>(P, nodes, tris, mats, lights, c, tracing, occluded, h, stack, lc);
        }
        HJR_TICK(2)
    }
#ifdef HJR_TIMING
    if (lane == 0) for (int i = 0; i < 3; i++) atomicAdd(&P.stats[HJR_NSTAT + i], tk[i]);
    if (__ffsll((long long)__ballot(true)) - 1 == (int)lane) for (int i = 0; i < 4; i++) atomicAdd(&P.stats[HJR_NSTAT + 3 + i], oc[i]);
    for (int i = 0; i < 10; i++) {
        unsigned long long v = td[i];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
        if (lane == 0 && v) atomicAdd(&P.stats[HJR_NSTAT + 7 + i], v);
    }
#endif

    if (STATS) {
        lc[10] = stack.n_over;
        for (int i = 0; i < HJR_NSTAT; i++) {
            unsigned long long v = lc[i];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
            if (lane == 0 && v) atomicAdd(&P.stats[i], v);
        }
    }
}
