// Workgroup-local wavefront form of the Henjou hot path (gfx950): the same per-lane bounce code as the megakernel
// (hjr_kernel.hip.h: bounce_pre_trace / bounce_post_trace), re-scheduled so that a wavefront only ever runs lanes that need the
// same thing.
//
// Why: in the megakernel a lane owns its path from the first to the last bounce, so each wave executes the union of what its 64
// lanes need — rays of very different length in one traversal loop, three material classes in one shading pass: 40 % of the VALU
// lanes do useful work while VALU issue is saturated (profiles/r02_*).  Here the 16 waves of the one workgroup per CU share a pool
// of `wf_cap` path contexts (LaneCtx, parked in HBM / Infinity Cache between stages: 128 bytes = one cache line per context, the pool
// of a CU is a few hundred KB) and take work in batches from five queues of 16-bit context ids in LDS:
//   * queue 0, TRACE: contexts with rays to trace (the pending NEE shadow ray of the bounce shaded last + the next closest-hit
//     ray, as in the megakernel's fused traversal).  A wave traces 64 of them; a lane whose rays are done notes the result and starts
//     the context it holds prefetched at once (lane-local), and every few passes the wave hands the finished contexts over and takes
//     new ones, so ray-length variance no longer idles lanes;
//   * queues 1..4, SHADE by the class of what the closest-hit ray found (path ends: miss / light; Disney; multiple-scattering
//     GGX; glass).  A wave takes up to 64 contexts of ONE class, runs bounce_post_trace + bounce_pre_trace on them (wave-uniform
//     branches in practice) and queues them for TRACE again.  The class only decides which lanes run together, never what a lane
//     computes: every pixel is bit-identical to the megakernel's and the oracle's.
// Contexts regenerate in place (a context whose path ends starts the next sample of its item, then the next item of the global
// queue), so the pool stays full until the frame runs out of work; a context with nothing left is retired, and the waves leave
// when the live count reaches zero.  All synchronisation is workgroup-local (LDS atomics + workgroup-scope fences): one
// workgroup is one CU, no cross-CU protocol is involved.
// What it buys (profiles/r02_experiments.md): 50 % active lanes and 20 % fewer VALU instructions than the megakernel, paid back in waits
// at the stage boundaries.  It wins where the shading pass is long — MIS: 187 vs 222 ms on the bundled scene, 417 vs 635 ms on 1 M
// triangles — and hjr_launch() selects it there; NEE and Pathtrace run faster on the megakernel.
#pragma once
#include "hjr_kernel.hip.h"

#ifndef HJR_WF_REFILL
#define HJR_WF_REFILL 16 /* trace stage: lanes without a ray (nothing prefetched either) before the wave stops for a hand-over */
#endif
#ifndef HJR_WF_PREFETCH_MIN
#define HJR_WF_PREFETCH_MIN 32 /* trace stage: lanes that have used up their prefetched context before the wave stops for a hand-over */
#endif
#ifndef HJR_WF_TRACE_MIN
#define HJR_WF_TRACE_MIN 32 /* scheduler: a TRACE batch is preferred over a partial SHADE batch from this many queued rays on */
#endif
#define HJR_WF_QUEUES 5

// LDS-typed pointers: accesses through them are ds_* instructions (lgkmcnt only).  A generic pointer would make them flat_*
// instructions, which also count in vmcnt and would make every queue operation wait for the trace stage's prefetch loads.
#define WF_LDS __attribute__((address_space(3)))
typedef WF_LDS volatile uint16_t* wf_ring_ptr;

// queue header in LDS (after the scene tables); rings of uint16 ids follow it.  Queue 0 (TRACE) has its own 32-bit counters.  The
// four SHADE queues share two 64-bit words per counter kind: word 0 = queue 1 | queue 2 << 32, word 1 = queue 3 | queue 4 << 32, so
// that a trace hand-over reserves (and publishes) its entries of two classes with ONE LDS atomic — "path ends" + Disney, which is all
// most hand-overs carry.  The 32-bit fields are free-running ring positions; a carry from the low into the high field would need 2^32
// pushes into one queue of one workgroup (hjr_device.hip keeps launches far below that).
struct WfShared {
    uint32_t head[HJR_WF_QUEUES]; // next ring position to take
    uint32_t live;                // contexts not yet retired
    uint32_t tail0, commit0;      // TRACE queue: next position to reserve / positions below this are written and may be taken
    unsigned long long tail_s[2];   // SHADE queues, packed pairs: next positions to reserve
    unsigned long long commit_s[2]; // SHADE queues, packed pairs: published positions (in reservation order)
    SharedRange items;            // the workgroup's range of the global work queue (hjr_kernel.hip.h)
};

static_assert(sizeof(WfShared) <= 96, "the kernel reserves 96 bytes of LDS for the queue header");

// context flags word (plane 1 .w)
#define WF_HAS_ITEM 1u
#define WF_DEAD 2u
#define WF_PATH_LIVE 4u
#define WF_FIN_PENDING 8u
#define WF_WRITE_PENDING 16u
#define WF_SH_VALID 32u
#define WF_FRESH 64u
#define WF_TRACING 128u /* the context has a closest-hit ray in this round (bounce_pre_trace's `tracing`) */
#define WF_MISS 0x7fffffffu

#ifdef HJR_WF_WATCHDOG
// diagnostic build: every waiting loop of the kernel gives up 1.5 s after the workgroup started (100 MHz real-time counter) and records where
__device__ unsigned int wf_where[8]; // [1] take, [2] push: slot wait, [3] push: publish wait, [4] trace loop, [5] pop, [6] scheduler
__shared__ unsigned long long wf_t0;
HD bool wf_expired(int where)
{
    if (__builtin_amdgcn_s_memrealtime() - wf_t0 < 150000000ull) return false;
    atomicAdd(&wf_where[where], 1u);
    return true;
}
#define WF_EXPIRED(w) wf_expired(w)
#else
#define WF_EXPIRED(w) false
#endif

#ifdef HJR_WF_TIMING
// diagnostic build: per-wave sums, flushed to wf_diag[] at the end: [0] scheduler idle clocks, [1] trace-stage clocks, [2] shade-stage clocks,
// [11] node-loop wave iterations, [12] lanes active in them, [13] triangle-loop wave iterations, [14] lanes active, [15] outer iterations, [16] lanes with a ray in them, [17] node-loop clocks, [18] leaf clocks
// [3] shade: clocks until the context loads have landed, [4] shade batches, [5] contexts in them, [6] trace hand-overs (report + refill), [7] rays handed
// over, [8] trace stage calls, [9] clocks inside hand-overs, [10] shade: clocks from the first store to the end of the pushes
__device__ unsigned long long wf_diag[24];
#define WF_T(i, expr) tdiag[i] += (expr)
#define WF_NOW() __builtin_amdgcn_s_memtime()
#else
#define WF_T(i, expr)
#define WF_NOW() 0ull
#endif

// ---- queue operations.  Every one is called by all 64 lanes of a wave at a wave-uniform point.
// Protocol: a producer reserves ring positions (tail), writes them, and PUBLISHES them in reservation order (commit); a taker
// claims published positions (head), so what it claims is always written and a taker never waits.  A slot holds id + 1 and is
// zeroed by its taker; a producer whose reserved slot still holds the entry of one lap ago (claimed, about to be read) waits for
// that zero.  Waiting is therefore one-directional — producers wait for takers and for earlier producers, takers for nobody —
// and cannot deadlock.  (An earlier form let takers claim committed COUNTS and wait for the slot: two producers one lap apart could
// then write the same slot, and the taker's wave-wide wait loop delayed its clears: lost entries and deadlocks under load.)
// Published position of queue q (acquire).
HD uint32_t wf_commit(WfShared* Q, int q, int order)
{
    if (q == 0) return order == __ATOMIC_ACQUIRE ? __hip_atomic_load(&Q->commit0, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) : __hip_atomic_load(&Q->commit0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const unsigned long long w = order == __ATOMIC_ACQUIRE ? __hip_atomic_load(&Q->commit_s[(q - 1) >> 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)
                                                           : __hip_atomic_load(&Q->commit_s[(q - 1) >> 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return (uint32_t)(w >> (((q - 1) & 1) * 32));
}
// Claims up to `want` published entries of queue q: returns how many (wave-uniform) and the first ring position.
HD uint32_t wf_pop(WfShared* Q, int q, uint32_t want, uint32_t& start)
{
    uint32_t got = 0, st = 0;
    if ((threadIdx.x & 63u) == 0u && want) {
        uint32_t h = __hip_atomic_load(&Q->head[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        for (;;) {
            const uint32_t avail = wf_commit(Q, q, __ATOMIC_ACQUIRE) - h;
            if (avail == 0u || avail > 0x7fffffffu || WF_EXPIRED(5)) break;
            const uint32_t take = avail < want ? avail : want;
            const uint32_t old = atomicCAS(&Q->head[q], h, h + take);
            if (old == h) { got = take; st = h; break; }
            h = old;
        }
    }
    start = (uint32_t)__builtin_amdgcn_readfirstlane((int)st);
    got = (uint32_t)__builtin_amdgcn_readfirstlane((int)got);
    if (got) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return got;
}
// The id at a claimed ring position (always written: see the protocol above); frees the slot.
HD uint32_t wf_take(wf_ring_ptr rings, int q, uint32_t pos, uint32_t cap)
{
    wf_ring_ptr slot = rings + (q * cap + (pos & (cap - 1u)));
    const uint32_t v = *slot;
    *slot = 0;
    return v - 1u;
}
// one ring slot: waits until the entry of one lap ago has been taken (a ring holds at most wf_cap ids: its taker zeroes it at once), then writes
HD void wf_put(wf_ring_ptr rings, int q, uint32_t pos, uint32_t id, uint32_t cap)
{
    wf_ring_ptr slot = rings + (q * cap + (pos & (cap - 1u)));
    while (*slot != 0) { if (WF_EXPIRED(2)) break; }
    *slot = (uint16_t)(id + 1u);
}
// Appends the ids of the lanes with `flag` to the TRACE queue.  Everything the wave wrote before (context records in memory, ring
// slots in LDS) is visible to the workgroup before the entries are published.
HD void wf_push_trace(WfShared* Q, wf_ring_ptr rings, bool flag, uint32_t id, uint32_t cap)
{
    const unsigned long long m = __ballot(flag);
    if (m == 0ull) return;
    const uint32_t n = (uint32_t)__popcll(m);
    const uint32_t prefix = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    uint32_t pos = 0;
    if ((threadIdx.x & 63u) == 0u) pos = atomicAdd(&Q->tail0, n);
    pos = (uint32_t)__builtin_amdgcn_readfirstlane((int)pos);
    if (flag) wf_put(rings, 0, pos + prefix, id, cap);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if ((threadIdx.x & 63u) == 0u) {
        while (__hip_atomic_load(&Q->commit0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != pos) { if (WF_EXPIRED(3)) break; } // earlier reservations publish first
        __hip_atomic_store(&Q->commit0, pos + n, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}
// Appends up to two entries per lane to the SHADE queues: e = (id + 1) | class << 16, 0 = none.  One reservation and one ordered
// publication per packed pair of queues that receives anything (see WfShared).
HD void wf_push_shade(WfShared* Q, wf_ring_ptr rings, uint32_t e0, uint32_t e1, uint32_t cap)
{
    if (__ballot(e0 != 0u) == 0ull) return; // (a lane fills e0 first)
    const uint32_t c0 = e0 >> 16, c1 = e1 >> 16;
    uint32_t n[4], pre0 = 0, pre1 = 0;
#pragma unroll
    for (uint32_t q = 0; q < 4u; q++) {
        const unsigned long long b0 = __ballot(e0 != 0u && c0 == q), b1 = __ballot(e1 != 0u && c1 == q);
        const uint32_t n0 = (uint32_t)__popcll(b0);
        n[q] = n0 + (uint32_t)__popcll(b1);
        if (c0 == q) pre0 = __builtin_amdgcn_mbcnt_hi((uint32_t)(b0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b0, 0u));
        if (c1 == q) pre1 = n0 + __builtin_amdgcn_mbcnt_hi((uint32_t)(b1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b1, 0u));
    }
    const bool lead = (threadIdx.x & 63u) == 0u;
    uint32_t base[4] = { 0u, 0u, 0u, 0u };
#pragma unroll
    for (int w = 0; w < 2; w++) {
        if (n[2 * w] + n[2 * w + 1] == 0u) continue; // wave-uniform
        unsigned long long old = 0ull;
        if (lead) old = atomicAdd(&Q->tail_s[w], (unsigned long long)n[2 * w] | ((unsigned long long)n[2 * w + 1] << 32));
        base[2 * w] = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)old);
        base[2 * w + 1] = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(old >> 32));
    }
    if (e0 != 0u) wf_put(rings, 1 + (int)c0, (c0 == 0u ? base[0] : (c0 == 1u ? base[1] : (c0 == 2u ? base[2] : base[3]))) + pre0, (e0 & 0xffffu) - 1u, cap);
    if (e1 != 0u) wf_put(rings, 1 + (int)c1, (c1 == 0u ? base[0] : (c1 == 1u ? base[1] : (c1 == 2u ? base[2] : base[3]))) + pre1, (e1 & 0xffffu) - 1u, cap);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lead) {
#pragma unroll
        for (int w = 0; w < 2; w++) {
            if (n[2 * w] + n[2 * w + 1] == 0u) continue;
            const unsigned long long mine = (unsigned long long)base[2 * w] | ((unsigned long long)base[2 * w + 1] << 32);
            while (__hip_atomic_load(&Q->commit_s[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != mine) { if (WF_EXPIRED(3)) break; } // earlier reservations publish first
            __hip_atomic_store(&Q->commit_s[w], (unsigned long long)(base[2 * w] + n[2 * w]) | ((unsigned long long)(base[2 * w + 1] + n[2 * w + 1]) << 32), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
}

// ---- context records: one per context id, HJR_WF_CTX_F4 float4 = 128 bytes = exactly one cache line.  A stage loads / stores a
// context with consecutive dwordx4 accesses of ONE line per lane; plane-major arrays (one line per 16-byte access) cost 8x the L2
// traffic and ran 1.7x slower.
//   0: ro.xyz rd.x   1: rd.yz sh_tmax flags   2: sh_d.xyz item   3: thr.xyz s   4: L.xyz mis.x   5: sumL.xyz (depth | rng_depth << 8 | it_cost << 20)
//   6: sh_contrib.xyz mis.y   7: mis.z | hit b1 b2 (k | occluded << 31)
// (mis = LaneCtx::mis_contrib; slot 7 .yzw is written by the TRACE stage, everything else by the SHADE stage)
// The albedo / normal sums of a launch with those AOVs live in a second array (HJR_WF_AOV_F4 float4 per context: sumA, sumN): they
// change once per SAMPLE (first hit: a read-modify-write there, LaneCtx::aov), the record is read and written once per BOUNCE by two
// stages — carrying them in the record (192 bytes, records straddling cache lines) cost 16 ms of 131 on the bundled scene.
#define HJR_MAX_DRAWS_PER_BOUNCE 26 /* 1 + 2 + 1 + 2 x 11, see wf_store_ctx */
#define HJR_WF_CTX_F4 8
#define HJR_WF_AOV_F4 2
HD void wf_store_ctx(float4* ctx, uint32_t id, const LaneCtx& c, bool tracing)
{
    const uint32_t flags = (c.has_item ? WF_HAS_ITEM : 0u) | (c.dead ? WF_DEAD : 0u) | (c.path_live ? WF_PATH_LIVE : 0u) | (c.fin_pending ? WF_FIN_PENDING : 0u) |
                           (c.write_pending ? WF_WRITE_PENDING : 0u) | (c.sh_valid ? WF_SH_VALID : 0u) | (c.fresh ? WF_FRESH : 0u) | (tracing ? WF_TRACING : 0u);
    float4* p = ctx + (size_t)id * HJR_WF_CTX_F4;
    p[0] = make_float4(c.ps.ro.x, c.ps.ro.y, c.ps.ro.z, c.ps.rd.x);
    p[1] = make_float4(c.ps.rd.y, c.ps.rd.z, c.sh_tmax, bits2f(flags));
    p[2] = make_float4(c.sh_d.x, c.sh_d.y, c.sh_d.z, bits2f(c.item));
    p[3] = make_float4(c.ps.thr.x, c.ps.thr.y, c.ps.thr.z, bits2f(c.s));
    p[4] = make_float4(c.ps.L.x, c.ps.L.y, c.ps.L.z, c.mis_contrib.x);
    // depth <= 10 (8 bits).  rng_depth (12 bits): CMJ draws of the path so far; a bounce draws at most HJR_MAX_DRAWS_PER_BOUNCE numbers (roulette 1,
    // light sample 2, the reference's discarded 2-D draw 1, a BSDF sample at most 11 — the multiple-scattering GGX walk: 6 heights + 5 normals —, twice
    // for MIS), a path at most 1 + 10 of those: checked below.  it_cost (12 bits) is a scheduling hint (closest-hit rays of the item so far,
    // hjr_cost_hist_kernel): it SATURATES instead of wrapping, so a long item can never look cheap.
    static_assert(1 + 10 * HJR_MAX_DRAWS_PER_BOUNCE < 4096, "rng_depth no longer fits its 12 bits of the context record");
    p[5] = make_float4(c.sumL.x, c.sumL.y, c.sumL.z, bits2f(((uint32_t)c.ps.depth & 0xffu) | ((c.ps.rng_depth & 0xfffu) << 8) | (min(c.it_cost, 0xfffu) << 20)));
    p[6] = make_float4(c.sh_contrib.x, c.sh_contrib.y, c.sh_contrib.z, c.mis_contrib.y);
    reinterpret_cast<float*>(p + 7)[0] = c.mis_contrib.z;
}
HD void wf_load_ctx(const float4* ctx, uint32_t id, LaneCtx& c, bool& tracing, float4& hitrec)
{
    const float4* p = ctx + (size_t)id * HJR_WF_CTX_F4;
    const float4 a = p[0], b = p[1], d = p[2], e = p[3], f = p[4], g = p[5], h = p[6];
    hitrec = p[7];
    const uint32_t flags = f2bits(b.w);
    c.has_item = flags & WF_HAS_ITEM; c.dead = flags & WF_DEAD; c.path_live = flags & WF_PATH_LIVE; c.fin_pending = flags & WF_FIN_PENDING;
    c.write_pending = flags & WF_WRITE_PENDING; c.sh_valid = flags & WF_SH_VALID; c.fresh = flags & WF_FRESH;
    tracing = flags & WF_TRACING;
    c.ps.ro = V(a.x, a.y, a.z); c.ps.rd = V(a.w, b.x, b.y); c.sh_tmax = b.z;
    c.sh_d = V(d.x, d.y, d.z); c.item = f2bits(d.w);
    c.ps.thr = V(e.x, e.y, e.z); c.s = f2bits(e.w);
    c.ps.L = V(f.x, f.y, f.z);
    c.sumL = V(g.x, g.y, g.z); c.ps.depth = (int)(f2bits(g.w) & 0xffu); c.ps.rng_depth = (f2bits(g.w) >> 8) & 0xfffu; c.it_cost = f2bits(g.w) >> 20;
    c.sh_contrib = V(h.x, h.y, h.z); c.mis_contrib = V(f.w, h.w, hitrec.x);
    c.sumA = V1(0.0f); c.sumN = V1(0.0f);
}


// ---- TRACE stage: the fused two-ray traversal of the megakernel (hjr_traverse.hip.h::traverse_fused: shadow ray, then the
// closest-hit ray, "while-while") with lane-level turnover.  phase: 0 shadow ray, 1 closest-hit ray, 2 no ray.
// Everything a lane does when its rays are done is LANE-LOCAL: it writes the hit into slot 7 of the context record, notes
// (context id, class of what was hit) in one of its two `fin` registers, and starts the context it holds prefetched (three float4 of
// the record, loaded one hand-over ahead) in the same iteration — no queue operation, no other lane involved.  Only the HAND-OVER is a
// wave-level step: all noted contexts go to the SHADE queues in one packed push (wf_push_shade) and every lane without a prefetched
// context takes one from the TRACE queue.  It runs when P.wf_prefetch_min lanes have used up their prefetched context (or
// P.wf_refill lanes have no ray at all), i.e. every few iterations instead of every iteration: rays of the bundled scene last 2.5
// iterations on average, and with the hand-over in every iteration it was 15 % of the kernel's time (profiles/r02_experiments.md).
// Returns when no lane has a ray, nothing is prefetched and the TRACE queue is empty.
template <bool STATS, int WIDTH, int BLOCK, typename ST>
HD void wf_trace_stage(const KParams& P, WfShared* Q, wf_ring_ptr rings, const float4* nodes, const float4* tris, const float4* mats, float4* ctx, ST& stack, unsigned long long* lc, unsigned long long* tdiag)
{
    constexpr int CTXF4 = HJR_WF_CTX_F4;
    const uint32_t cap = P.wf_cap;
    const float tmin = 0.001f;
    const f3 cam_o = V(P.cam_pos[0], P.cam_pos[1], P.cam_pos[2]);
    int phase = 2, sp = 0;
    uint32_t cur = HJR_TRAV_DONE, id = 0;
    f3 o = V1(0.0f), d = V(1.0f, 0.0f, 0.0f), ro = V1(0.0f), db = V1(0.0f);
    float a_tmax = 0.0f;
    bool b_valid = false, fresh = false, occluded = false;
    Hit hit; hit.prim = 0xffffffffu; hit.t = 1e16f; hit.k = 0; hit.b1 = hit.b2 = 0.0f;
    BoxRay<WIDTH, ST::kNodesInLds> R = box_ray<WIDTH, ST::kNodesInLds>(nodes, o, d);
    bool n_valid = false; // prefetched context
    uint32_t n_id = 0;
    float4 n0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), n1 = n0, n2 = n0;
    uint32_t fin0 = 0u, fin1 = 0u; // finished contexts not handed over yet: (id + 1) | class << 16, 0 = none
    // a lane's rays are done: hit record -> slot 7 of the context record (.x of the slot belongs to the SHADE stage), context noted for the hand-over
    auto finish = [&]() {
        uint32_t kk = WF_MISS, cls = 0u;
        if (hit.prim != 0xffffffffu) {
            kk = hit.k;
            const float4* m = mats + f2bits(tris[hit.k * HJR_TRI_F4 + 2].z) * HJR_MAT_F4;
            const float4 m0 = m[0], m3 = m[3];
            cls = f2bits(m3.x) != 0 ? 0u : (f2bits(m3.y) != 0 ? 3u : (m0.w > 0.5f ? 2u : 1u)); // light | glass | metallic (msGGX) | Disney
        }
        float* hp = reinterpret_cast<float*>(ctx + (size_t)id * CTXF4 + 7) + 1;
        hp[0] = hit.b1; hp[1] = hit.b2; hp[2] = bits2f(kk | (occluded ? 0x80000000u : 0u));
        const uint32_t e = (id + 1u) | (cls << 16);
        if (fin0 == 0u) fin0 = e; else fin1 = e;
        phase = 2; cur = HJR_TRAV_DONE;
    };
    for (;;) {
        if (WF_EXPIRED(4)) return;
        // ---- a lane without a ray starts its prefetched context (its loads were issued at least one hand-over ago)
        if (phase == 2 && n_valid && fin1 == 0u) {
            const uint32_t flags = f2bits(n1.w);
            id = n_id;
            ro = V(n0.x, n0.y, n0.z); db = V(n0.w, n1.x, n1.y); a_tmax = n1.z;
            b_valid = flags & WF_TRACING; fresh = flags & WF_FRESH;
            occluded = false;
            phase = (flags & WF_SH_VALID) ? 0 : 1; // a queued context has at least one of the two rays
            hit.prim = 0xffffffffu; hit.t = (phase == 0) ? a_tmax : 1e16f; // hit.t = far end of the ray being traced (as in traverse_fused)
            o = (phase == 0 || !fresh) ? ro : cam_o;
            d = (phase == 0) ? V(n2.x, n2.y, n2.z) : db;
            R = box_ray<WIDTH, ST::kNodesInLds>(nodes, o, d);
            sp = 0; cur = 0;
            n_valid = false;
        }
        const uint32_t n_idle = (uint32_t)__popcll(__ballot(phase == 2));
        const unsigned long long m_need = __ballot(!n_valid);
        if (n_idle >= P.wf_refill || (uint32_t)__popcll(m_need) >= P.wf_prefetch_min) {
            [[maybe_unused]] const unsigned long long t_h0 = WF_NOW();
            WF_T(6, 1); WF_T(7, __popcll(__ballot(fin0 != 0u)) + __popcll(__ballot(fin1 != 0u)));
            wf_push_shade(Q, rings, fin0, fin1, cap);
            fin0 = fin1 = 0u;
            uint32_t start = 0;
            const uint32_t got = wf_pop(Q, 0, (uint32_t)__popcll(m_need), start);
            if (got) {
                const uint32_t prefix = __builtin_amdgcn_mbcnt_hi((uint32_t)(m_need >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m_need, 0u));
                if (!n_valid && prefix < got) {
                    n_id = wf_take(rings, 0, start + prefix, cap);
                    const float4* cp = ctx + (size_t)n_id * CTXF4;
                    n0 = cp[0]; n1 = cp[1]; n2 = cp[2];
                    n_valid = true;
                }
            }
            WF_T(9, WF_NOW() - t_h0);
            if (__ballot(phase < 2 || n_valid) == 0ull) return;
        }
        WF_T(15, 1); WF_T(16, __popcll(__ballot(phase < 2)));
        if (phase < 2) {
            for (;;) { // every lane first descends through inner nodes until it holds a leaf (or is out of work) ...
                if (!(cur & HJR_LEAF_FLAG)) {
                    WF_T(11, 1); WF_T(12, __popcll(__ballot(true)));
                    const uint32_t nb = node_step<WIDTH, BLOCK, ST>(nodes, cur, R, tmin, hit.t, stack, sp);
                    if (STATS) { if (phase == 0) lc[5] += nb; else lc[3] += nb; }
                }
                // ... or until fewer than P.node_min lanes are still descending: those keep their node for the next pass
                const uint32_t n_inner = (uint32_t)__popcll(__ballot(!(cur & HJR_LEAF_FLAG)));
                if (n_inner == 0u || n_inner < P.node_min) break;
            }
            bool done = (cur == HJR_TRAV_DONE); // ... then all lanes that hold a leaf test its triangles together
            if (!done && (cur & HJR_LEAF_FLAG)) {
                const uint32_t first = cur & 0x07ffffffu, count = (cur >> 27) & 15u;
                const float tri_tmax = (phase == 0) ? a_tmax : 1e16f;
                for (uint32_t i = 0; i < count; i++) {
                    WF_T(13, 1); WF_T(14, __popcll(__ballot(true)));
                    const float4* g = tris + (first + i) * HJR_TRI_F4;
                    const float4 g0 = g[0], g1 = g[1], g2 = g[2];
                    float t, b1, b2;
                    if (STATS) { if (phase == 0) lc[6] += 1; else lc[4] += 1; }
                    if (ray_tri<ST::kTriEarly>(V(g0.x, g0.y, g0.z), V(g0.w, g1.x, g1.y), V(g1.z, g1.w, g2.x), o, d, tmin, tri_tmax, t, b1, b2)) {
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
                if (STATS) { if (phase == 0) lc[2] += 1; else lc[1] += 1; }
                if (phase == 0 && b_valid) { // this lane's shadow ray is resolved: start its closest-hit ray right away
                    phase = 1;
                    hit.t = 1e16f;
                    o = fresh ? cam_o : ro; d = db;
                    R = box_ray<WIDTH, ST::kNodesInLds>(nodes, o, d);
                    sp = 0; cur = 0;
                } else finish();
            }
        }
    }
}

// ---- SHADE stage: up to 64 contexts of one class: second half of the bounce just traced, first half of the next one
template <int INTEGRATOR, bool STATS, bool AOVS, bool TEX, int WIDTH, int BLOCK, typename ST>
HD void wf_shade_stage(const KParams& P, WfShared* Q, wf_ring_ptr rings, int q, const float4* nodes, const float4* tris, const float4* mats, const float4* lights,
                       float4* ctx, WaveRange& wr, ST& stack, unsigned long long* lc, unsigned long long* tdiag)
{
    [[maybe_unused]] const unsigned long long t_s0 = WF_NOW();
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t cap = P.wf_cap;
    uint32_t start = 0;
    const uint32_t got = wf_pop(Q, q, 64u, start);
    if (got == 0u) return;
    const bool have = lane < got;
    LaneCtx c;
    ctx_reset(c);
    c.dead = true; // lanes without a context take no part in the refill
    bool tracing = false;
    uint32_t id = 0;
    if (have) {
        id = wf_take(rings, q, start + lane, cap);
        float4 hr;
        wf_load_ctx(ctx, id, c, tracing, hr);
        c.ps.rng_head = path_head(P, HJR_PX(c), HJR_PY(c), c.s); // not in the record: rebuilt once per pass, used by both halves of the bounce
        if (AOVS) c.aov = P.wf_aov + ((size_t)blockIdx.x * cap + id) * HJR_WF_AOV_F4; // the item's albedo / normal sums live here, not in the record
#ifdef HJR_WF_TIMING
        __builtin_amdgcn_s_waitcnt(0x0070); // vmcnt(0): the loads have landed
        if (lane == 0u) { WF_T(3, WF_NOW() - t_s0); }
#endif
        const uint32_t kk = f2bits(hr.w);
        Hit h;
        h.t = 0.0f; h.b1 = hr.y; h.b2 = hr.z; h.k = kk & 0x7fffffffu; // (the hit distance is not an input of the hit program)
        h.prim = (h.k == WF_MISS) ? 0xffffffffu : f2bits(tris[h.k * HJR_TRI_F4 + 2].y);
        bounce_post_trace<INTEGRATOR, STATS, AOVS, TEX, WIDTH, BLOCK, ST, true>(P, nodes, tris, mats, lights, c, tracing, (kk >> 31) != 0u, h, stack, lc);
    }
    tracing = false;
    bounce_pre_trace<STATS, AOVS, true>(P, c, wr, have, tracing, lc); // (ONEWRITE: one write-out site per pass)
    const bool again = have && (tracing || c.sh_valid);
    // no ray, not dead: the item it took lies outside a ragged frame edge; it takes the next one in another pass (class "path ends")
    const bool retry = have && !again && !c.dead;
    [[maybe_unused]] const unsigned long long t_st = WF_NOW();
    WF_T(4, 1); WF_T(5, got);
    if (again || retry) wf_store_ctx(ctx, id, c, tracing);
    wf_push_trace(Q, rings, again, id, cap);
    wf_push_shade(Q, rings, retry ? id + 1u : 0u, 0u, cap); // class 0
    const uint32_t retired = (uint32_t)__popcll(__ballot(have && !again && !retry)); // no ray and no item left: the context is finished
    if (retired && lane == 0u) atomicSub(&Q->live, retired);
    WF_T(10, WF_NOW() - t_st);
}

// Dynamic LDS: [traversal stacks: stack_lds_entries x BLOCK uint32][scene tables when LDSBVH][WfShared][rings: HJR_WF_QUEUES x wf_cap uint16]
// SPILL: only the top stack_lds_entries of a lane's traversal stack are in LDS, deeper ones in the HBM overflow buffer (always for the
// memory layouts; for the LDS-resident layout only when the whole stacks do not fit beside the scene tables and the queues)
template <int INTEGRATOR, bool STATS, int BLOCK, bool LDSBVH, bool SPILL, int WIDTH, int VAR>
__global__ void __launch_bounds__(BLOCK, 1) hjr_wavefront_kernel(const KParams P)
{
    constexpr bool AOVS = VAR >= 1, TEX = VAR == 2; // as in hjr_render_kernel
    typedef uint32_t SE;
    typedef LaneStack<SE, BLOCK, SPILL, STATS, LDSBVH, false> ST; // (last: the branch-free shape of the triangle test)
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
    float4* after_stacks = hjr_smem + (BLOCK * P.stack_lds_entries * (uint32_t)sizeof(SE) + 15u) / 16u;
    const uint32_t scene_f4 = LDSBVH ? (P.n_node_f4 + P.n_tri_f4 + P.n_mat_f4 + P.n_light_f4) : 0u;
    WfShared* Q = reinterpret_cast<WfShared*>(after_stacks + scene_f4);
    wf_ring_ptr rings = (wf_ring_ptr)(after_stacks + scene_f4 + 6); // the header takes 96 bytes
    const uint32_t cap = P.wf_cap;
    // all contexts start in SHADE queue 1 (class "path ends") with every flag clear: their first pass does nothing but take an item
    for (uint32_t i = threadIdx.x; i < HJR_WF_QUEUES * cap; i += BLOCK) rings[i] = (i >= cap && i < 2u * cap) ? (uint16_t)(i - cap + 1u) : (uint16_t)0;
    if (threadIdx.x < HJR_WF_QUEUES) Q->head[threadIdx.x] = 0u;
    if (threadIdx.x == 0u) {
        Q->tail0 = Q->commit0 = 0u;
        Q->tail_s[0] = Q->commit_s[0] = (unsigned long long)cap; Q->tail_s[1] = Q->commit_s[1] = 0ull; // queue 1 holds every context
        Q->live = cap; Q->items.range = 0ull; Q->items.lock = 0u; Q->items.exhausted = 0u;
    }
#ifdef HJR_WF_WATCHDOG
    if (threadIdx.x == 0u) wf_t0 = __builtin_amdgcn_s_memrealtime();
#endif
    constexpr int CTXF4 = HJR_WF_CTX_F4;
    float4* ctx = P.wf_ctx + (size_t)blockIdx.x * cap * CTXF4;
    for (uint32_t i = threadIdx.x; i < cap * CTXF4; i += BLOCK) // every context starts with all flags clear but `fresh`
        ctx[i] = (i % CTXF4 == 1u) ? make_float4(0.0f, 0.0f, 0.0f, bits2f(WF_FRESH)) : ((i % CTXF4 == 7u) ? make_float4(0.0f, 0.0f, 0.0f, bits2f(WF_MISS)) : make_float4(0.0f, 0.0f, 0.0f, 0.0f));
    if (LDSBVH) stage_scene_in_lds<SE, BLOCK>(P, after_stacks, nodes, tris, mats, lights); // ends with a barrier
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

    unsigned long long lc[HJR_NSTAT];
    if (STATS) for (int i = 0; i < HJR_NSTAT; i++) lc[i] = 0;
    WaveRange wr; wr.next = wr.end = 0u; wr.exhausted = false; wr.shared = &Q->items; wr.base_item = 0u;
#ifdef HJR_WF_TIMING
    unsigned long long tdiag[19] = { 0 };
#else
    unsigned long long* tdiag = nullptr;
#endif

    for (;;) {
        // wave-uniform choice of the next batch: a full SHADE batch first (largest class), then TRACE, then whatever is there
        uint32_t pick = 7u; // 0 trace, 1..4 shade class, 6 leave, 7 wait
        if (lane == 0u) {
            // published entries per queue (commit is read before head, so the difference can only err on the low side; clamp the rest)
            auto queued = [&](uint32_t q) {
                const uint32_t cm = wf_commit(Q, (int)q, __ATOMIC_RELAXED);
                const uint32_t d = cm - __hip_atomic_load(&Q->head[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                return d > 0x7fffffffu ? 0u : d;
            };
            const uint32_t c0 = queued(0u);
            uint32_t best = 0u, bq = 1u;
            for (uint32_t q = 1; q < HJR_WF_QUEUES; q++) {
                const uint32_t cq = queued(q);
                if (cq > best) { best = cq; bq = q; }
            }
            if (best >= 64u) pick = bq;
            else if (c0 >= P.wf_trace_min) pick = 0u;
            else if (best > 0u && best >= c0) pick = bq;
            else if (c0 > 0u) pick = 0u;
            else if (__hip_atomic_load(&Q->live, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0u) pick = 6u;
        }
        pick = (uint32_t)__builtin_amdgcn_readfirstlane((int)pick);
#ifdef HJR_WF_WATCHDOG
        if (pick != 6u && WF_EXPIRED(6)) { // diagnostic build: record the queue state of the first workgroup that runs out of time
            if (lane == 0u && atomicAdd(&P.stats[HJR_NSTAT], 1ull) == 0ull) {
                for (int q = 0; q < HJR_WF_QUEUES; q++) {
                    P.stats[HJR_NSTAT + 1 + q] = wf_commit(Q, q, __ATOMIC_RELAXED); P.stats[HJR_NSTAT + 6 + q] = Q->head[q];
                    P.stats[HJR_NSTAT + 11 + q] = q == 0 ? Q->tail0 : (uint32_t)(Q->tail_s[(q - 1) >> 1] >> (((q - 1) & 1) * 32));
                }
                P.stats[HJR_NSTAT + 16] = Q->live; P.stats[HJR_NSTAT + 17] = blockIdx.x; P.stats[HJR_NSTAT + 18] = (uint32_t)(Q->items.range >> 32) - (uint32_t)Q->items.range;
            }
            break;
        }
#endif
        if (pick == 6u) break;
        if (pick == 7u) {
#ifdef HJR_WF_TIMING
            const unsigned long long t_i0 = WF_NOW();
#endif
            __builtin_amdgcn_s_sleep(16);
            WF_T(0, WF_NOW() - t_i0);
            continue;
        }
        [[maybe_unused]] const unsigned long long t_g0 = WF_NOW();

        if (pick == 0u) { wf_trace_stage<STATS, WIDTH, BLOCK, ST>(P, Q, rings, nodes, tris, mats, ctx, stack, lc, tdiag); WF_T(8, 1); WF_T(1, WF_NOW() - t_g0); }
        else { wf_shade_stage<INTEGRATOR, STATS, AOVS, TEX, WIDTH, BLOCK, ST>(P, Q, rings, (int)pick, nodes, tris, mats, lights, ctx, wr, stack, lc, tdiag); WF_T(2, WF_NOW() - t_g0); }
    }
#ifdef HJR_WF_TIMING
    if (lane == 0u) for (int i = 0; i < 19; i++) atomicAdd(&wf_diag[i], tdiag[i]);
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
