// Software ray traversal (replaces optixTrace + the RT cores): canonical ray / triangle test, per-lane stacks, BVH2 / BVH4 node
// steps, stand-alone traversal and the fused two-ray traversal with straggler carry-over.
#pragma once
#include "hjr_params.hip.h"

#ifdef HJR_FAST_MATH /* the traversal and the triangle test are the same operations in every build: no mul + add fusion here */
#pragma clang fp contract(off)
#endif

// ------------------------------------------------------------------ traversal: replaces optixTrace (kernel/rt.h:15-69) + RT cores
struct Counters {
    uint32_t box, tri;
};

HD float dotf(f3 a, f3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
HD f3 crossf(f3 a, f3 b)
{
    return V(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x)));
}
// canonical ray/triangle test — the same operation sequence as the oracle's ray_tri().  Two shapes of the same arithmetic:
// with early returns (EARLY; the megakernel on LDS-resident scenes: whole waves leave at the first two tests often enough, 2.5 % faster there) and
// with all seven conditions evaluated and combined at the end (scenes read from memory and the wavefront kernels: the fewer branches the better,
// 1.2 % / 0.8 % faster there).  A zero
// determinant makes inv infinite and u infinite or NaN; `det != 0` keeps the result explicit.  (profiles/r03_experiments.md)
template <bool EARLY>
HD bool ray_tri(f3 v0, f3 v1, f3 v2, f3 o, f3 d, float tmin, float tmax, float& t, float& b1, float& b2)
{
    f3 e1 = v1 - v0, e2 = v2 - v0;
    f3 p = crossf(d, e2);
    float det = dotf(e1, p);
    if (EARLY && det == 0.0f) return false;
    float inv = exact_div(1.0f, det); // == 1.0f / det, whatever the division flags of the translation unit
    f3 tv = o - v0;
    float u = dotf(tv, p) * inv;
    if (EARLY && !(u >= 0.0f && u <= 1.0f)) return false;
    f3 q = crossf(tv, e1);
    float v = dotf(d, q) * inv;
    if (EARLY && !(v >= 0.0f && u + v <= 1.0f)) return false;
    float tt = dotf(e2, q) * inv;
    const bool ok = (det != 0.0f) & (u >= 0.0f) & (u <= 1.0f) & (v >= 0.0f) & (u + v <= 1.0f) & (tt > tmin) & (tt < tmax);
    if (ok) { t = tt; b1 = u; b2 = v; }
    return ok;
}

// ---- per-lane traversal stack in LDS, element i of this lane at stack[i * BLOCK] (conflict-free columns).  Small scenes that
// are staged into LDS use 16-bit entries (node index < 32768, or leaf: bit15 | count << 13 | first triangle < 8192), which
// halves the stack's LDS footprint; everything else uses the 32-bit child refs as they are.
// The 16-bit form holds leaves of at most HJR_STACK16_LEAF_MAX triangles (2-bit count): host/frame.cpp only selects it
// for such trees, and hjr_selftest_stack16 (csrc/hjr_device.hip) round-trips every ref the builder can emit.
#define HJR_STACK16_LEAF_MAX 3u
#define HJR_STACK16_MAX_TRIS 8192u
#define HJR_STACK16_MAX_NODES 32768u
#define HDH __host__ __device__ __forceinline__
template <typename ST> HDH ST stack_enc(uint32_t ref);
template <> HDH uint32_t stack_enc<uint32_t>(uint32_t ref) { return ref; }
template <> HDH uint16_t stack_enc<uint16_t>(uint32_t ref)
{
    return (uint16_t)((ref & HJR_LEAF_FLAG) ? (0x8000u | (((ref >> 27) & 3u) << 13) | (ref & 0x1fffu)) : ref);
}
HDH uint32_t stack_dec(uint32_t r) { return r; }
HDH uint32_t stack_dec(uint16_t r16)
{
    const uint32_t r = r16;
    return (r & 0x8000u) ? (HJR_LEAF_FLAG | (((r >> 13) & 3u) << 27) | (r & 0x1fffu)) : r;
}

// One lane's traversal stack.  SPILL == false: every entry in LDS (column of this lane).  SPILL == true (kernels that read the
// BVH from memory): only the top-of-tree `lds_n` entries are in LDS, deeper ones overflow into a per-lane column of a global
// buffer ([level][lane], coalesced when neighbouring lanes overflow together).  The exact worst-case depth of a BVH4 over a
// million triangles is ~46 entries, traversal rarely needs more than a dozen: with the whole stack in LDS the stacks, not the
// registers, capped the occupancy at 3 workgroups per CU.  lds_n is a launch parameter (wave-uniform, an SGPR compare):
// HJR_SHORT_STACK entries unless option "short_stack" overrides it (tests force the overflow path with 2).
#ifndef HJR_SHORT_STACK
#define HJR_SHORT_STACK 16
#endif
template <typename E, int BLOCK_, bool SPILL, bool COUNT = false, bool NLDS = false, bool TRI_EARLY = NLDS>
struct LaneStack {
    static constexpr bool kTriEarly = TRI_EARLY; // which shape of ray_tri the kernel uses: early returns pay in the LDS-resident megakernel only
    static constexpr bool kNodesInLds = NLDS; // the kernel stages nodes (and triangles) in LDS: node addresses are 32-bit LDS addresses
    static constexpr bool kSpill = SPILL;
    // memory layouts (SPILL): the first n_top nodes of the breadth-first BVH4 — the top of the tree, which every ray walks — are also staged in
    // LDS by the workgroup (`top`); node_step reads a node from there when its id is below n_top (0: no copy)
    const float4* top;
    uint32_t n_top;
    E* lds;
    uint32_t* spill;
    uint32_t spill_stride;
    int lds_n;
    uint32_t n_over; // COUNT (the counting kernel variant) only: pushes that went to the overflow buffer
    // The overflow accesses are written as inline assembly with their own wait: the compiler's wait-count pass does not see them, so
    // it puts no `s_waitcnt vmcnt(0)` at the merge points of the traversal loop for a load that almost never happens (with ordinary
    // loads every iteration of the LDS-resident node loop waited for whatever else the wave had in flight, e.g. prefetched rays).
    // A lane only reads back what it stored itself (same wave, same address: returned in order).
    HD void put(int i, uint32_t ref)
    {
        if (!SPILL || i < lds_n) lds[i * BLOCK_] = stack_enc<E>(ref);
        else {
            uint32_t* a = spill + (size_t)(i - lds_n) * spill_stride;
            asm volatile("global_store_dword %0, %1, off" : : "v"(a), "v"(ref) : "memory");
            if (COUNT) n_over++;
        }
    }
    // n more entries on top of sp fit the LDS part of the column (whole-stack layouts: always — the builder sizes the column for the worst case)
    HD bool room_for(int sp, int n) const { return !SPILL || sp + n <= lds_n; }
    HD void put_lds(int i, uint32_t ref) { lds[i * BLOCK_] = stack_enc<E>(ref); }
    HD uint32_t get(int i) const
    {
        if (!SPILL || i < lds_n) return stack_dec(lds[i * BLOCK_]);
        const uint32_t* a = spill + (size_t)(i - lds_n) * spill_stride;
        uint32_t r;
        asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(a) : "memory");
        return r;
    }
};

// ---- box-test side of a ray.  The slab test only has to be conservative (boxes are padded): it uses the
// 1-ulp hardware reciprocal and (plane - o) * inv evaluated as fma(plane, inv, -o * inv).  Direction components smaller than
// 1e-30 are clamped (sign kept) so that inv stays finite and no inf - inf can appear for axis-parallel rays.
typedef float v2f __attribute__((ext_vector_type(2))); // 8-byte loads of a plane pair
typedef uint32_t v2u __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const char lds_cchar;
HD float box_dir(float d) { return (fabsf(d) < 1e-30f) ? copysignf(1e-30f, d) : d; }
// BVH4 (nodes always read from memory): reciprocal direction, -o * inv, and the near-row selectors (1: the component is negative)
// BVH2: the same six floats, and per axis the address of the ray's near-plane pair in node 0 — node base + the byte offset the
// direction's sign selects (hjr_layout.h) — as a 32-bit LDS address when the nodes are staged in LDS (NLDS).
template <int WIDTH, bool NLDS> struct BoxRay {
    f3 inv, oi;
    uint32_t sx, sy, sz;
};
template <bool NLDS> struct BoxRay<2, NLDS> {
    f3 inv, oi;
    typename std::conditional<NLDS, uint32_t, const char*>::type ax, ay, az;
};
template <int WIDTH, bool NLDS> HD BoxRay<WIDTH, NLDS> box_ray(const float4* nodes, f3 o, f3 d)
{
    BoxRay<WIDTH, NLDS> r;
    const f3 dd = V(box_dir(d.x), box_dir(d.y), box_dir(d.z));
    const f3 inv = V(__builtin_amdgcn_rcpf(dd.x), __builtin_amdgcn_rcpf(dd.y), __builtin_amdgcn_rcpf(dd.z));
    const f3 oi = V(-o.x * inv.x, -o.y * inv.y, -o.z * inv.z);
    if constexpr (WIDTH == 2) {
        r.inv = inv; r.oi = oi;
        const uint32_t kx = dd.x < 0.0f ? 8u : 0u, ky = dd.y < 0.0f ? 24u : 16u, kz = dd.z < 0.0f ? 40u : 32u;
        if constexpr (NLDS) {
            const uint32_t base = (uint32_t)reinterpret_cast<uintptr_t>((lds_cchar*)reinterpret_cast<const char*>(nodes));
            r.ax = base + kx; r.ay = base + ky; r.az = base + kz;
        } else {
            const char* base = reinterpret_cast<const char*>(nodes);
            r.ax = base + kx; r.ay = base + ky; r.az = base + kz;
        }
    } else {
        r.inv = inv; r.oi = oi;
        r.sx = dd.x < 0.0f ? 1u : 0u; r.sy = dd.y < 0.0f ? 1u : 0u; r.sz = dd.z < 0.0f ? 1u : 0u;
    }
    return r;
}
// the 8-byte plane pair / child pair at byte address a (+ an immediate)
template <typename T> HD T node_ld(uint32_t a, uint32_t imm) { return *reinterpret_cast<__attribute__((address_space(3))) const T*>((uintptr_t)(a + imm)); }
template <typename T> HD T node_ld(const char* a, uint32_t imm) { return *reinterpret_cast<const T*>(a + imm); }
HD uint32_t node_far(uint32_t a) { return a ^ 8u; }
HD const char* node_far(const char* a) { return reinterpret_cast<const char*>(reinterpret_cast<uintptr_t>(a) ^ (uintptr_t)8); }

#define HJR_TRAV_DONE 0xffffffffu
// One inner-node step: tests the children of node `cur` against [tmin, tfar], continues with the nearest hit child, pushes
// the other hit children, or pops (HJR_TRAV_DONE when the stack is empty).  Returns the number of boxes tested.
template <int WIDTH, int BLOCK, typename ST>
HD uint32_t node_step(const float4* nodes, uint32_t& cur, const BoxRay<WIDTH, ST::kNodesInLds>& R, float tmin, float tfar, ST& stack, int& sp)
{
    if constexpr (WIDTH == 2) {
    // A ray reads the near and the far plane pair of each axis straight from the node (rows of (lo0 lo1 hi0 hi1), the pair picked by
    // the sign of the direction: an address, not a min / max) and folds the three axes with max3 / min3: 7 address + 12 fma + 8 min /
    // max instructions, against 12 fma + 18 min / max for the lo / hi box layout of rounds 1 - 2 (bundled scene: 125.7 -> 122.4 ms).
    // Nodes are 64-byte aligned, so the far pair of an axis is at (near address ^ 8).  One v_pk_fma_f32 per pair instead of two
    // v_fma_f32 was measured 2.5 % SLOWER (packed fp32 fma does not issue at the scalar rate here; profiles/r03_experiments.md).
    const uint32_t nofs = cur * (uint32_t)(HJR_NODE2_F4 * 16);
    const auto anx = R.ax + nofs, any = R.ay + nofs, anz = R.az + nofs;
    const v2f nx = node_ld<v2f>(anx, 0), fx = node_ld<v2f>(node_far(anx), 0);
    const v2f ny = node_ld<v2f>(any, 0), fy = node_ld<v2f>(node_far(any), 0);
    const v2f nz = node_ld<v2f>(anz, 0), fz = node_ld<v2f>(node_far(anz), 0);
    const v2u cc = node_ld<v2u>(anx, 48); // the child refs are stored twice (bytes 48 and 56): one immediate offset from the near-x address
    const f3 inv = R.inv, oi = R.oi;
    // min(t, tfar) spelled as the instruction: fminf() puts a quieting copy of tfar (v_max_f32 t, t) in front of it in every step of this loop,
    // because the compiler cannot see that hit.t is never a signalling NaN; the operands here never are NaNs at all (finite planes, clamped directions)
    float fz0 = fmaf(fz.x, inv.z, oi.z), fz1 = fmaf(fz.y, inv.z, oi.z);
    asm("v_min_f32 %0, %1, %2" : "=v"(fz0) : "v"(fz0), "v"(tfar));
    asm("v_min_f32 %0, %1, %2" : "=v"(fz1) : "v"(fz1), "v"(tfar));
    const float lo0 = fmaxf(fmaxf(fmaf(nx.x, inv.x, oi.x), fmaf(ny.x, inv.y, oi.y)), fmaxf(fmaf(nz.x, inv.z, oi.z), tmin));
    const float hi0 = fminf(fminf(fmaf(fx.x, inv.x, oi.x), fmaf(fy.x, inv.y, oi.y)), fz0);
    const float lo1 = fmaxf(fmaxf(fmaf(nx.y, inv.x, oi.x), fmaf(ny.y, inv.y, oi.y)), fmaxf(fmaf(nz.y, inv.z, oi.z), tmin));
    const float hi1 = fminf(fminf(fmaf(fx.y, inv.x, oi.x), fmaf(fy.y, inv.y, oi.y)), fz1);
    const bool h0 = lo0 <= hi0, h1 = lo1 <= hi1; // conservative through the 2^-15 box padding (>= 16x the rounding error of t)
    const uint32_t c0 = cc.x, c1 = cc.y;
    // (the push stays a branch here: writing the farther child unconditionally and advancing the top by 0 / 1, as the BVH4 step below does,
    // was measured slower for the BVH2 layouts — megakernel 118.8 -> 120.1 ms, MIS on the wavefront kernel 176.2 -> 179.6: the LDS pipe is the busier one)
    if (h0 && h1) {
        const bool swap = lo1 < lo0;
        stack.put(sp, swap ? c0 : c1);
        sp++;
        cur = swap ? c1 : c0;
    } else if (h0) cur = c0;
    else if (h1) cur = c1;
    else if (sp > 0) { sp--; cur = stack.get(sp); }
    else cur = HJR_TRAV_DONE;
    return 2u;
    } else {
    const float4* nd = nodes + cur * HJR_NODE4_F4;
    if constexpr (ST::kSpill) { if (cur < stack.n_top) nd = stack.top + cur * HJR_NODE4_F4; } // top of the tree: LDS copy (flat loads serve both)
    const f3 inv = R.inv, oi = R.oi;
    const float INF = bits2f(0x7f800000u);
    // near / far plane rows picked by the ray's direction signs: no min/max per axis
    const float4 nx = nd[0 + R.sx], fx = nd[1 - R.sx];
    const float4 ny = nd[2 + R.sy], fy = nd[3 - R.sy];
    const float4 nz = nd[4 + R.sz], fz = nd[5 - R.sz];
    const float4 rr = nd[6];
#define HJR_CHILD(c, C)                                                                                                  \
    float tn##C = fmaxf(fmaxf(fmaf(nx.c, inv.x, oi.x), fmaf(ny.c, inv.y, oi.y)), fmaxf(fmaf(nz.c, inv.z, oi.z), tmin)); \
    const float tf##C = fminf(fminf(fmaf(fx.c, inv.x, oi.x), fmaf(fy.c, inv.y, oi.y)), fminf(fmaf(fz.c, inv.z, oi.z), tfar)); \
    const bool h##C = tn##C <= tf##C;                                                                                   \
    tn##C = h##C ? tn##C : INF;
    HJR_CHILD(x, 0) HJR_CHILD(y, 1) HJR_CHILD(z, 2) HJR_CHILD(w, 3)
#undef HJR_CHILD
    const uint32_t r0 = f2bits(rr.x), r1 = f2bits(rr.y), r2 = f2bits(rr.z), r3 = f2bits(rr.w);
    const float m = fminf(fminf(tn0, tn1), fminf(tn2, tn3));
    // nearest hit child first (ties: lowest slot); the other hit children are pushed in slot order
    const bool e0 = tn0 == m, e1 = (tn1 == m) & !e0, e2 = (tn2 == m) & !e0 & !e1, e3 = !e0 & !e1 & !e2; // one-hot: the selected child
    const bool p3 = h3 & !e3, p2 = h2 & !e2, p1 = h1 & !e1, p0 = h0 & !e0;
    if (stack.room_for(sp, 4)) { // the usual case: four slots of this lane's LDS column are free: write each candidate at the running top
        stack.put_lds(sp, r3); sp += p3 ? 1 : 0; // and advance only for those that are pushed (a slot written in vain is overwritten by the
        stack.put_lds(sp, r2); sp += p2 ? 1 : 0; // next one): no branch, no overflow test per push
        stack.put_lds(sp, r1); sp += p1 ? 1 : 0;
        stack.put_lds(sp, r0); sp += p0 ? 1 : 0;
    } else {
        if (p3) { stack.put(sp, r3); sp++; }
        if (p2) { stack.put(sp, r2); sp++; }
        if (p1) { stack.put(sp, r1); sp++; }
        if (p0) { stack.put(sp, r0); sp++; }
    }
    if (h0 | h1 | h2 | h3) cur = e0 ? r0 : (e1 ? r1 : (e2 ? r2 : r3));
    else if (sp > 0) { sp--; cur = stack.get(sp); }
    else cur = HJR_TRAV_DONE;
    return 4u;
    }
}

struct Hit { float t, b1, b2; uint32_t k, prim; };

// stack: this lane's column of the LDS stack, element i at stack[i * BLOCK]
template <bool ANY, bool STATS, int WIDTH, int BLOCK, typename ST>
HD bool traverse(const float4* nodes, const float4* tris, f3 o, f3 d, float tmin, float tmax, Hit& hit, ST& stack, Counters& cnt)
{
    BoxRay<WIDTH, ST::kNodesInLds> R = box_ray<WIDTH, ST::kNodesInLds>(nodes, o, d);
    int sp = 0;
    uint32_t cur = 0;
    hit.prim = 0xffffffffu;
    hit.t = tmax;
    for (;;) {
        while (!(cur & HJR_LEAF_FLAG)) { // descend through inner nodes until this lane holds a leaf (or is done)
            const uint32_t nb = node_step<WIDTH, BLOCK, ST>(nodes, cur, R, tmin, hit.t, stack, sp);
            if (STATS) cnt.box += nb;
        }
        if (cur == HJR_TRAV_DONE) break;
        const uint32_t first = cur & 0x07ffffffu, count = (cur >> 27) & 15u;
        for (uint32_t i = 0; i < count; i++) {
            const float4* g = tris + (first + i) * HJR_TRI_F4;
            const float4 g0 = g[0], g1 = g[1], g2 = g[2];
            float t, b1, b2;
            if (STATS) cnt.tri++;
            if (ray_tri<ST::kTriEarly>(V(g0.x, g0.y, g0.z), V(g0.w, g1.x, g1.y), V(g1.z, g1.w, g2.x), o, d, tmin, tmax, t, b1, b2)) {
                if (ANY) return true;
                const uint32_t prim = f2bits(g2.y);
                // order-independent closest-hit rule: smaller t wins; equal t -> smaller global prim id
                if (hit.prim == 0xffffffffu || t < hit.t || (t == hit.t && prim < hit.prim)) {
                    hit.t = t; hit.b1 = b1; hit.b2 = b2; hit.k = first + i; hit.prim = prim;
                }
            }
        }
        if (sp == 0) break;
        sp--;
        cur = stack.get(sp);
    }
    return hit.prim != 0xffffffffu;
}

// Fused traversal of two rays per lane in ONE loop: ray A = the pending NEE shadow ray of the bounce just shaded (any-hit),
// ray B = the next closest-hit ray (continuation or a regenerated primary ray).  A lane moves on to B the moment its A is
// resolved, without waiting for the rest of the wave, so the wave's trip count is max_lanes(tripsA + tripsB) instead of
// max(tripsA) + max(tripsB) — the SIMT cost of per-lane trip-count variance drops by ~1/3 (profiles/r01_experiments.md).
// Results are identical to two separate traversals.
//
// Lane threshold of the descent loop (node_min; also in hjr_wavefront.hip.h::wf_trace_stage): the lanes descend together until every
// lane holds a leaf — or until fewer than node_min lanes are still descending.  Those few keep their inner node and carry on in the
// next pass of the outer loop, next to the lanes that have meanwhile finished their leaf, instead of holding the whole wave idle for
// their descent (1 M triangles: 28 descent iterations per pass with 18 of 64 lanes working before; 277 -> 179 ms with node_min = 24;
// bundled scene 134 -> 129 ms with 4).  Which triangles a lane tests, and in which order, does not change.  (The stand-alone
// traverse() above keeps the plain loop: with the threshold MIS ran 3 % slower on the bundled scene.)
//
// Straggler carry-over (CARRY > 0): the loop also ends when at most CARRY lanes are still traversing (and at least one
// lane of this round has finished).  Those lanes keep their traversal state (TravCarry + hit + their LDS stack column),
// skip the shading that follows and resume in the next round next to the other lanes' new rays: the wave's trip count per
// round is set by the (64 - CARRY)-th slowest lane instead of the slowest one.  Per-lane results do not change.  The
// threshold trades traversal lane-occupancy against shading lane-occupancy (profiles/r01_experiments.md): 14 for the
// LDS-resident scenes (8 in rounds 1 - 2; with the shading code of round 3 — shorter by its zero-weighted terms — 8 / 10 / 12 / 14 /
// 16 / 18 / 20 lanes give 111.35 / 110.5 / 110.05 / 110.0 / 110.25 / 110.65 / 111.4 ms on the headline launch with node_min 4, and 14 lanes
// with node_min 6 - 7 109.1), 32 when nodes come
// from memory (traversal-heavy).
#ifndef HJR_CARRY_LDS
#define HJR_CARRY_LDS 14
#endif
#ifndef HJR_CARRY_MEM
#define HJR_CARRY_MEM 32
#endif
struct TravCarry { uint32_t cur; int sp, phase; };
template <bool STATS, int WIDTH, int BLOCK, typename ST, int CARRY>
HD bool traverse_fused(const float4* nodes, const float4* tris, const bool a_valid, const f3 ao, const f3 ad, const float a_tmax, const bool b_valid,
                       const f3 bo, const f3 bd, bool& occluded, Hit& hit, ST& stack, Counters& ca, Counters& cb, const bool resume, TravCarry& tc, const uint32_t node_min,
                       unsigned long long* td = nullptr)
{
#ifdef HJR_TIMING /* diagnostic build: lane occupancy of the loop's parts.  A wave-level event is counted by the first active lane, lane sums by every lane */
#define HJR_TD_WAVE(i) { if (__ffsll((long long)__ballot(true)) - 1 == (int)(threadIdx.x & 63u)) td[i] += 1; }
#define HJR_TD_LANE(i) { td[i] += 1; }
#else
#define HJR_TD_WAVE(i) {}
#define HJR_TD_LANE(i) {}
#endif
    const float tmin = 0.001f;
    int phase, sp;
    uint32_t cur;
    if (CARRY > 0 && resume) { phase = tc.phase; sp = tc.sp; cur = tc.cur; } // occluded / hit are the caller's, kept across rounds
    else {
        occluded = false;
        hit.prim = 0xffffffffu;
        hit.t = a_valid ? a_tmax : 1e16f; // hit.t is the far end of the ray being traced: the shadow ray's tmax in phase 0, the closest hit so far in phase 1
        phase = a_valid ? 0 : (b_valid ? 1 : 2);
        sp = 0;
        cur = (phase < 2) ? 0u : HJR_TRAV_DONE;
    }
    f3 o = (phase == 0) ? ao : bo;
    f3 d = (phase == 0) ? ad : bd;
    BoxRay<WIDTH, ST::kNodesInLds> R = box_ray<WIDTH, ST::kNodesInLds>(nodes, o, d);
    const int n_start = CARRY > 0 ? __popcll(__ballot(phase < 2)) : 0;
    for (;;) {
        if (CARRY > 0) {
            const int n_act = __popcll(__ballot(phase < 2));
            if (n_act == 0 || (n_act <= CARRY && n_act < n_start)) break;
        } else if (__ballot(phase < 2) == 0ull) break;
        HJR_TD_WAVE(0)
        if (phase < 2) {
        HJR_TD_LANE(1)
        if (phase == 0) { HJR_TD_LANE(8) }
        // "while-while" traversal: every lane first descends through inner nodes until it holds a leaf (or is out of work), or until
        // fewer than node_min lanes are still descending (they go on in the next pass) ...
        for (;;) {
            HJR_TD_WAVE(2)
            if (!(cur & HJR_LEAF_FLAG)) {
                HJR_TD_LANE(3)
                const uint32_t nb = node_step<WIDTH, BLOCK, ST>(nodes, cur, R, tmin, hit.t, stack, sp);
                if (STATS) { if (phase == 0) ca.box += nb; else cb.box += nb; }
            }
            const uint32_t n_inner = (uint32_t)__popcll(__ballot(!(cur & HJR_LEAF_FLAG)));
            if (n_inner == 0u || n_inner < node_min) break;
        }
        // ... then all lanes that hold a leaf test its triangles together
        bool done = (cur == HJR_TRAV_DONE);
        if (!done && (cur & HJR_LEAF_FLAG)) {
            HJR_TD_WAVE(4)
            HJR_TD_LANE(5)
            const uint32_t first = cur & 0x07ffffffu, count = (cur >> 27) & 15u;
            const float tri_tmax = (phase == 0) ? a_tmax : 1e16f;
            for (uint32_t i = 0; i < count; i++) {
                HJR_TD_WAVE(6)
                HJR_TD_LANE(7)
                const float4* g = tris + (first + i) * HJR_TRI_F4;
                const float4 g0 = g[0], g1 = g[1], g2 = g[2];
                float t, b1, b2;
                if (STATS) { if (phase == 0) ca.tri++; else cb.tri++; }
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
            if (phase == 0 && b_valid) { // this lane's shadow ray is resolved: start its closest-hit ray right away
                phase = 1;
                hit.t = 1e16f;
                o = bo; d = bd;
                R = box_ray<WIDTH, ST::kNodesInLds>(nodes, o, d);
                sp = 0; cur = 0;
            } else { phase = 2; cur = HJR_TRAV_DONE; }
        }
    } }
    if (CARRY > 0) { tc.phase = phase; tc.sp = sp; tc.cur = cur; return phase < 2; }
    return false;
}

#ifdef HJR_FAST_MATH
#pragma clang fp contract(fast)
#endif
