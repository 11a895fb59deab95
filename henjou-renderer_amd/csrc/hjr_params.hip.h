// Kernel parameter block of the Henjou hot path: the scalar part of the reference's Params (kernel/Params.h, filled at
// renderer/renderer.h:1175-1227) plus the device arrays of the flattened scene (csrc/hjr_layout.h).
#pragma once
#include <type_traits>

#include "hjr_layout.h"
#include "hjr_math.hip.h"

struct KParams {
    const float4* nodes;
    const float4* tri_geom;
    const float4* tri_shade;
    const uint32_t* tri_inst;
    const float4* materials;
    const float4* lights;
    const uchar4* lut;
    const uchar4* texels;    // RGBA8 atlas of all material textures
    const uint4* tex_desc;   // per texture slot: (texel offset, width, height, srgb)
    const float* srgb_lut;   // 256-entry sRGB -> linear table (host-computed)
    const float4* sky_tex;   // equirect IBL (float4), null = constant sky
    float4* aov_color;
    float4* aov_albedo;
    float4* aov_normal;
    unsigned int* queue_head;
    unsigned long long* stats;
    unsigned long long* nan_list;    // counting launches: [0] NaN / Inf samples seen, [1 ..] the first HJR_NAN_LIST of them as x | y << 13 | sample << 26
    int lut_w, lut_h;
    int sky_w, sky_h;
    uint32_t n_lights;
    uint32_t width, height, spp, frame, seed, integrator;
    uint32_t tiles_x, n_owned_items; // items = owned tiles * n_chunks * 64
    uint32_t rank, world;
    uint32_t packed;                 // HJR_FLAG_PACKED: the AOV buffers hold this rank's tiles only, [owned tile][64] float4
    uint32_t chunk_spp, n_chunks;    // samples per work item, work items per pixel (hjr_chunking)
    uint32_t n_node_f4, n_tri_f4;    // float4 counts of nodes[] / tri_geom[] (LDS staging)
    uint32_t n_mat_f4, n_light_f4;   // float4 counts of materials[] / lights[] (staged behind the triangles in the LDS variant)
    uint32_t stack_depth;            // traversal stack entries per lane (BVH depth + 1)
    uint32_t* stack_spill;           // memory-path kernels: overflow of the short LDS stacks, [level][lane]
    const uint32_t* tile_order;      // owned tiles, expensive first (hjr_classify_tiles_kernel); null = plain round-robin order
    uint32_t* tile_order_w;          // the same buffer, writable (pre-pass kernels)
    uint32_t* tile_class;            // per owned tile: costliest first hit of its pixel centres: 0 background / light, 1 Disney, 2 metallic (msGGX), 3 glass
    uint32_t* tile_count;            // [0..3] tiles per class, [4..7] scatter cursors
    uint32_t n_owned_tiles;
    uint32_t* tile_bucket;           // per owned tile: sort key of the measured-cost order
    uint32_t* tile_cost;             // per owned tile: closest-hit rays traced for it this frame (feeds the next frame's tile order)
    uint32_t* cost_hist;             // [0..63] tiles per cost bucket, [64..127] scatter cursors
    uint32_t cost_div;               // 64 * spp: rays per tile at one ray per sample
    uint32_t spill_stride;           // lanes in the grid
    uint32_t n_top_nodes;            // memory-path megakernel, BVH4: nodes [0, n_top_nodes) are also staged in LDS by every workgroup (0: none)
    uint32_t stack_lds_entries;      // memory-path kernels: stack entries per lane kept in LDS (the rest overflow to stack_spill)
    uint32_t node_min;               // descent loop of the fused traversals: lanes still holding an inner node below which a pass moves on to the leaves (hjr_traverse.hip.h)
    uint32_t hold_min, hold_age;     // megakernel: hits of the rare material class wait until a wave has hold_min of them or one has waited hold_age rounds (0: off)
    float4* wf_ctx;                  // wavefront kernel: context records, [workgroup][wf_cap] x 128 bytes (hjr_wavefront.hip.h)
    float4* wf_aov;                  // wavefront kernel, albedo / normal launches: per-context AOV sums, [workgroup][wf_cap] x 32 bytes
    uint32_t wf_cap;                 // contexts per workgroup (power of two, <= 32768: ids travel as uint16 + 1)
    uint32_t wf_refill, wf_trace_min; // trace-stage hand-over threshold (lanes without a ray) / scheduler preference for TRACE (queued rays)
    uint32_t wf_prefetch_min;        // trace-stage hand-over threshold (lanes that have used up their prefetched context)
    float4* part_color;              // [n_chunks][owned tile][64] chunk sums when n_chunks > 1
    float4* part_albedo;
    float4* part_normal;
    float cam_pos[3], cam_dir[3], cam_up[3], cam_right[3];
    float cam_f;
    float sky[3]; // scene_sky_default * ibl_intensity
    float ibl_intensity;
};
