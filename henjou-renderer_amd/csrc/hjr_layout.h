// Device data layout shared by the host-side frame builder and the HIP kernel.
// Everything is 16-byte records so that each per-lane fetch is a global_load_dwordx4.
#pragma once
#include <stdint.h>

#define HJR_TILE 8u                 /* 8x8 pixel tiles = one wavefront of pixels */
/* Tile ids of the pixel-tile shard.  Tile (tx, ty) has id  t = ty * tiles_x + (tx + ty) % tiles_x : rows of tiles, row ty rotated by ty
 * places.  Tile t belongs to rank t % world and is that rank's (t / world)-th tile.  Without the rotation a frame whose tiles_x is a
 * multiple of the GPU count (1080p: 240, 4K: 480; 2, 4, 8 GPUs) would give every rank a fixed set of 8-pixel vertical stripes; with it
 * a rank's tiles run along diagonals, whatever tiles_x is.  Per-rank tile counts are unchanged (each row is a permutation of itself). */
#if defined(__HIPCC__) || defined(__CUDACC__)
#define HJR_LAYOUT_FN __host__ __device__ static inline
#else
#define HJR_LAYOUT_FN static inline
#endif
HJR_LAYOUT_FN uint32_t hjr_tile_id(uint32_t tx, uint32_t ty, uint32_t tiles_x) { return ty * tiles_x + (tx + ty) % tiles_x; }
HJR_LAYOUT_FN void hjr_tile_xy(uint32_t t, uint32_t tiles_x, uint32_t* tx, uint32_t* ty)
{
    const uint32_t y = t / tiles_x, c = t - y * tiles_x, r = y % tiles_x;
    *ty = y;
    *tx = c >= r ? c - r : c + tiles_x - r;
}
#define HJR_STACK_DEPTH 32          /* per-lane traversal stack entries (LDS); the builder caps tree depth at this */
#define HJR_LEAF_FLAG 0x80000000u   /* child ref: bit31 = leaf, bits 27..30 = triangle count, bits 0..26 = first triangle */
#define HJR_LEAF_MAX 4u              /* encoding cap */
#define HJR_LEAF_DEFAULT 2u          /* builder default: measured fastest on MI355X (profiles/r01_experiments.md) */
#define HJR_MAX_TRIS (1u << 27)

/* Two node formats; the host picks per frame (host/frame.cpp):
 *  - BVH2 when the whole tree + triangles fit into LDS beside the traversal stacks (small scenes): fewest VALU
 *    instructions per ray, and LDS hides the extra dependent steps;
 *  - BVH4 (the BVH2 collapsed) otherwise: half the dependent memory round trips when nodes come from L2 / HBM.
 * Measured on MI355X (profiles/r01_experiments.md): cornelbox in LDS 223 ms (BVH2) vs 249 ms (BVH4); 1 M-triangle stress scene
 * from memory 643 ms (BVH2) vs 501 ms (BVH4).
 *
 * BVH2 node, 64 B = 4 x float4; holds the (padded) boxes of both children, one row per axis with the children side by side:
 *   q0 = (lo0.x lo1.x hi0.x hi1.x)  q1 = (lo0.y lo1.y hi0.y hi1.y)  q2 = (lo0.z lo1.z hi0.z hi1.z)
 *   q3 = (child0, child1, child0, child1) as uint bits
 *   A ray reads the near pair and the far pair of each axis (8 bytes each) at an offset picked by the sign of its direction,
 *   so the slab test of both children is 6 packed fmas + max3 / min3, no per-axis min / max (hjr_traverse.hip.h::node_step).
 * BVH4 node, 112 B = 7 x float4, child-major planes so that one 16-byte read yields the same plane of all four children:
 *   q0 = lo.x[0..3]  q1 = hi.x[0..3]  q2 = lo.y[0..3]  q3 = hi.y[0..3]  q4 = lo.z[0..3]  q5 = hi.z[0..3]  q6 = child refs[0..3]
 *   A ray picks its near/far plane rows by the sign of its direction, so the slab test needs no min/max.
 *   Unused child slots: inverted box (+1e30 / -1e30) and an empty-leaf ref. */
#ifndef HJR_NODE2_F4
#define HJR_NODE2_F4 4
#endif
#define HJR_NODE4_F4 7
#ifndef HJR_BLOCK_LDS
#define HJR_BLOCK_LDS 1024          /* threads of the one-per-CU workgroup that shares an LDS copy of the BVH (16 waves = 4 per SIMD) */
#endif
#define HJR_LDS_BUDGET (159u * 1024u)
/* Triangle (leaf order), 48 B = 3 x float4: world-space vertices + global prim id.
 *   g0 = (v0.x v0.y v0.z v1.x)  g1 = (v1.y v1.z v2.x v2.y)  g2 = (v2.z, prim_id bits, material_id bits, 0) */
#define HJR_TRI_F4 3
/* Shading record by GLOBAL prim id, 64 B = 4 x float4: world normals (each normalised, __closesthit__ch) + uvs + material.
 *   s0 = (n0.xyz uv0.x) s1 = (n1.xyz uv0.y) s2 = (n2.xyz uv1.x) s3 = (uv1.y uv2.x uv2.y material_id bits) */
#define HJR_SHADE_F4 4
/* Material, 80 B = hjr_material verbatim (include/henjou_hip.h); m4 = (metallic_roughness_tex, normal_tex, emission_tex, -) */
#define HJR_MAT_F4 5
/* Light triangle, 96 B = 6 x float4 (light_sample.h:43-72 hoisted to once per frame):
 *   l0 = (v0.xyz pdf)  l1 = (v1.xyz em.x)  l2 = (v2.xyz em.y)  l3 = (n0.xyz em.z)  l4 = (n1.xyz prim)  l5 = (n2.xyz 1.0f / pdf)
 *   v*: transform_position(transforms[inst]); n*: transform_normal(inv_transforms[inst]) (NOT normalised);
 *   pdf = float(1.0 / area) * (1.0f / light_prim_count) */
#define HJR_LIGHT_F4 6

#define HJR_NAN_LIST 8 /* NaN / Inf samples a counting launch locates (hjr_stats.nan_where) */
#define HJR_NSTAT 11 /* hjr_stats' ten leading uint64 counters in order, then [10] = stack_overflow_pushes */

/* Work-item chunking: a pixel's spp samples are cut into n_chunks runs of chunk_spp consecutive samples
 * (a multiple of 8, at most 64 runs); pixel mean = ((c0 + c1) + ... ) * (1/spp), ck = in-order sum of run k.  Depends on spp
 * only, so results do not depend on scheduling, tile sharding or GPU count.  A run is one work item, executed by one lane
 * from its first to its last bounce: its length is the critical path of a launch (16-sample runs: ~4 ms for a pixel inside
 * the glass sphere, 20 % of an 8-GPU share of the C2 frame), hence runs of 8 (profiles/r01_experiments.md). */
static inline uint32_t hjr_chunk_spp(uint32_t spp) { uint32_t n8 = (spp + 7u) / 8u; return 8u * ((n8 + 63u) / 64u); }
static inline uint32_t hjr_n_chunks(uint32_t spp) { uint32_t s = hjr_chunk_spp(spp); return (spp + s - 1u) / s; }
