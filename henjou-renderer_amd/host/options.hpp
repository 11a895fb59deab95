// Tuning / test options of a device context (include/henjou_hip.h: hjr_set_option / hjr_get_option).  The default build of the library reads
// NO environment variable: behaviour depends on its arguments only.  Experiment builds (`make variant`, -DHJR_ENV_OPTIONS) additionally
// take every option from the environment at hjr_create (HJR_<KEY IN UPPER CASE>), so that one binary can be swept from a shell script.
#pragma once
#include <cstdint>
#include <cstring>

namespace hjr {

enum Opt {
    OPT_PIPELINE, OPT_LDS_BVH, OPT_LDS_STACK16, OPT_BVH_WIDTH, OPT_LEAF_MAX, OPT_NODE_MIN, OPT_HOLD_MIN, OPT_HOLD_AGE, OPT_SHORT_STACK,
    OPT_BLOCKS_PER_CU, OPT_TILE_ORDER, OPT_WF_CAP, OPT_WF_REFILL, OPT_WF_PREFETCH_MIN, OPT_WF_TRACE_MIN, OPT_HOST_THREADS, OPT_VERBOSE,
    OPT_FORCE_REBUILD, OPT_TOP_NODES, OPT_BVH_REFINE, OPT_COUNT
};
struct OptDesc { const char* key; int lo, hi; };
// value -1 always means "the library's default"; the ranges are those of explicit values
inline const OptDesc* opt_table()
{
    static const OptDesc t[OPT_COUNT] = {
        { "pipeline", 0, 2 },        // 0 per launch what was measured faster (wavefront kernels for MIS, megakernel otherwise), 1 megakernel, 2 wavefront kernel
        { "lds_bvh", 0, 1 },         // 1 stage BVH2 + triangles in LDS when they fit (default), 0 always read the scene from memory       [next hjr_set_transforms]
        { "lds_stack16", 0, 1 },     // 1 prefer 16-bit LDS stack entries whenever the tree admits them (default: only when 32-bit ones do not fit) [next hjr_set_transforms]
        { "bvh_width", 2, 4 },       // 2 / 4: force the node format (4 also forces the memory path); 3 is rejected                       [next hjr_set_transforms]
        { "leaf_max", 1, 4 },        // triangles per BVH leaf (default 2)                                                              [next hjr_set_transforms]
        { "node_min", 1, 64 },       // descent loops: lanes still descending below which a pass moves on to the leaves (default 6 / 8 / 24 by layout and kernel family)
        { "hold_min", 0, 64 },       // megakernel, LDS layouts: metallic hits a wave collects before it shades them (0 never holds; default 8)
        { "hold_age", 1, 1000 },     // ... or rounds the oldest of them has waited (default 2)
        { "short_stack", 1, 64 },    // memory layouts: traversal-stack entries per lane kept in LDS (default 16; deeper ones overflow to HBM)
        { "blocks_per_cu", 1, 8 },   // memory layouts: workgroups per CU of the persistent grid (default: what the occupancy query returns)
        { "tile_order", 0, 2 },      // 0 plain tile order, 1 first-hit classes, 2 classes + measured cost of the previous frame (default 1 at N = 1, 2 at N > 1)
        { "wf_cap", 64, 32768 },     // wavefront kernel: path contexts per workgroup, a power of two (default 2048 / 4096)
        { "wf_refill", 1, 64 },      // wavefront kernel: hand-over threshold of the trace stage, lanes without a ray
        { "wf_prefetch_min", 1, 64 },// ... lanes that have used up their prefetched context
        { "wf_trace_min", 1, 4096 }, // wavefront kernel: queued rays from which the scheduler prefers TRACE
        { "host_threads", 1, 256 },  // worker threads of the per-frame host preparation (default min(hardware threads, 16)); process-wide
        { "verbose", 0, 1 },         // 1: BVH format, sizes and host build time per frame on stderr
        { "force_rebuild", 0, 1 },   // 1: rebuild the frame data even when the transforms did not change (benchmarking)
        { "top_nodes", 0, 1024 },    // memory layouts, BVH4: nodes of the top of the tree every workgroup also stages in LDS (0 none)
        { "bvh_refine", 0, 16 },     // insertion-based refinement passes over the built BVH2 (default 0 up to 65536 triangles, 1 above)  [next hjr_set_transforms]
    };
    return t;
}
inline int opt_find(const char* key)
{
    if (!key) return -1;
    const OptDesc* t = opt_table();
    for (int i = 0; i < OPT_COUNT; i++) if (strcmp(t[i].key, key) == 0) return i;
    return -1;
}
struct Options {
    int v[OPT_COUNT];
    Options() { for (int& x : v) x = -1; }
    int get(Opt o, int def) const { return v[o] < 0 ? def : v[o]; }
    bool is_set(Opt o) const { return v[o] >= 0; }
};
// what host/frame.cpp needs of them
struct BuildOptions {
    bool allow_lds = true, prefer_stack16 = false, timing = false;
    int bvh_width = -1, leaf_max = -1, refine = -1;
};
void set_host_threads(int n); // host/frame.cpp; n <= 0 restores the default

} // namespace hjr
