#include "frame.hpp"

#include <algorithm>
#include <cstdio>
#include <chrono>
#include <atomic>
#include <functional>
#include <thread>
#include <limits>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>

namespace hjr {

bool SceneCopy::set(const hjr_scene_view& v, std::string& err)
{
    if (v.n_triangles >= HJR_MAX_TRIS) { err = "too many triangles"; return false; }
    if (v.n_triangles && (!v.vertices || !v.normals || !v.texcoords || !v.indices || !v.material_ids)) { err = "null geometry pointer"; return false; }
    if (v.n_instances && !v.prim_offset) { err = "null prim_offset"; return false; }
    if (v.n_materials && !v.materials) { err = "null materials"; return false; }
    if (v.n_lights && (!v.light_prim_ids || !v.light_prim_emission)) { err = "null light arrays"; return false; }
    if (v.n_triangles && v.n_instances == 0) { err = "triangles without instances"; return false; }
    n_triangles = v.n_triangles;
    n_instances = v.n_instances;
    vertices.assign(v.vertices, v.vertices + 3 * (size_t)v.n_vertices);
    normals.assign(v.normals, v.normals + 3 * (size_t)v.n_vertices);
    texcoords.assign(v.texcoords, v.texcoords + 2 * (size_t)v.n_vertices);
    indices.assign(v.indices, v.indices + 3 * (size_t)v.n_triangles);
    material_ids.assign(v.material_ids, v.material_ids + v.n_triangles);
    prim_offset.assign(v.prim_offset, v.prim_offset + v.n_instances);
    materials.assign(v.materials, v.materials + v.n_materials);
    light_prim_ids.assign(v.light_prim_ids, v.light_prim_ids + v.n_lights);
    light_prim_emission.assign(v.light_prim_emission, v.light_prim_emission + 3 * (size_t)v.n_lights);
    for (uint32_t i = 0; i < 3 * v.n_triangles; i++)
        if (indices[i] >= v.n_vertices) { err = "vertex index out of range"; return false; }
    for (uint32_t t = 0; t < v.n_triangles; t++)
        if (material_ids[t] >= v.n_materials) { err = "material id out of range"; return false; }
    for (uint32_t i = 0; i < v.n_instances; i++)
        if (prim_offset[i] > v.n_triangles || (i && prim_offset[i] < prim_offset[i - 1])) { err = "prim_offset not ascending"; return false; }
    if (v.n_instances && prim_offset[0] != 0) { err = "prim_offset[0] must be 0"; return false; }
    for (uint32_t l = 0; l < v.n_lights; l++)
        if (light_prim_ids[l] >= v.n_triangles) { err = "light prim id out of range"; return false; }
    textures.clear(); texels.clear();
    if (v.n_textures && !v.textures) { err = "null texture array"; return false; }
    for (uint32_t t = 0; t < v.n_textures; t++) {
        const hjr_texture& tx = v.textures[t];
        if (!tx.rgba8 || tx.width == 0 || tx.height == 0 || (uint64_t)tx.width * tx.height > (1u << 28)) { err = "bad texture"; return false; }
        Tex d = { (uint32_t)texels.size(), tx.width, tx.height, tx.srgb };
        const uint32_t* px = reinterpret_cast<const uint32_t*>(tx.rgba8);
        texels.insert(texels.end(), px, px + (size_t)tx.width * tx.height);
        textures.push_back(d);
    }
    for (uint32_t m = 0; m < v.n_materials; m++) {
        const hjr_material& mt = materials[m];
        if (mt.basecolor_tex >= (int)v.n_textures || mt.metallic_roughness_tex >= (int)v.n_textures || mt.normal_tex >= (int)v.n_textures) { err = "material texture slot out of range"; return false; }
    }
    return true;
}

namespace {

// Host worker threads for the per-frame scene preparation (flatten, BVH build, emit).  Every parallel loop below is either
// element-wise or merges per-chunk partial results that are exact (min / max / integer counts), so the emitted arrays do not
// depend on the thread count.  Default min(hardware threads, 16); hjr::set_host_threads (option "host_threads") overrides it process-wide.
std::atomic<int> g_host_threads{ 0 };
inline unsigned host_threads()
{
    const int forced = g_host_threads.load(std::memory_order_relaxed);
    if (forced >= 1) return (unsigned)std::min(forced, 256);
    static const unsigned n = [] {
        unsigned h = std::thread::hardware_concurrency();
        return h == 0 ? 1u : std::min(h, 16u);
    }();
    return n;
}
// f(begin, end, chunk): [0, n) cut into `chunks` equal ranges (a pure function of n and grain), run on host_threads() threads
template <class F> void parallel_chunks(size_t n, size_t grain, F f)
{
    const size_t chunks = std::max<size_t>(1, std::min<size_t>(64, n / std::max<size_t>(grain, 1)));
    auto range = [&](size_t c, size_t& b, size_t& e) { b = n * c / chunks; e = n * (c + 1) / chunks; };
    const unsigned T = (unsigned)std::min<size_t>(host_threads(), chunks);
    if (T <= 1) { for (size_t c = 0; c < chunks; c++) { size_t b, e; range(c, b, e); f(b, e, c); } return; }
    std::atomic<size_t> next(0);
    auto work = [&] { for (;;) { size_t c = next.fetch_add(1); if (c >= chunks) return; size_t b, e; range(c, b, e); f(b, e, c); } };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < T; t++) th.emplace_back(work);
    work();
    for (auto& t : th) t.join();
}
inline size_t n_chunks_of(size_t n, size_t grain) { return std::max<size_t>(1, std::min<size_t>(64, n / std::max<size_t>(grain, 1))); }

struct V3 { float x, y, z; };
inline V3 sub(V3 a, V3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; }
// kernel/math.h:73-87 with sutil's dot(float4,float4) = x+y+z+w terms left to right
inline V3 transform_position(const float* m, V3 p)
{
    return { m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3] * 1.0f, m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7] * 1.0f,
             m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11] * 1.0f };
}
inline V3 transform_normal(const float* m, V3 n)
{
    return { m[0] * n.x + m[4] * n.y + m[8] * n.z + 0.0f * 0.0f, m[1] * n.x + m[5] * n.y + m[9] * n.z + 0.0f * 0.0f,
             m[2] * n.x + m[6] * n.y + m[10] * n.z + 0.0f * 0.0f };
}
inline V3 normalize(V3 v)
{
    float inv = 1.0f / sqrtf(dot(v, v));
    return { v.x * inv, v.y * inv, v.z * inv };
}
inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

struct Box {
    float lo[3], hi[3];
    void reset() { for (int a = 0; a < 3; a++) { lo[a] = FLT_MAX; hi[a] = -FLT_MAX; } }
    void grow(const Box& b) { for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], b.lo[a]); hi[a] = std::max(hi[a], b.hi[a]); } }
    void grow(const float* p) { for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], p[a]); hi[a] = std::max(hi[a], p[a]); } }
    float area() const
    {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return (dx < 0) ? 0.0f : 2.0f * (dx * dy + dy * dz + dz * dx);
    }
};

struct BuildNode { Box box; int left = -1, right = -1; uint32_t first = 0, count = 0; };

struct BuildTask { int node; uint32_t first, count, depth; };

struct Builder {
    uint32_t leaf_max = HJR_LEAF_DEFAULT;
    const Box* tbox = nullptr;   // per triangle (padded)
    const float* cent = nullptr; // 3 per triangle
    uint32_t* order = nullptr;   // permutation being partitioned in place (subtrees own disjoint ranges)
    std::vector<BuildNode> nodes;
    uint32_t max_depth = 0;
    // Large scenes: subtrees of at most `defer_below` triangles are not built by the top-level recursion but recorded as tasks
    // (their root node is a placeholder) and built afterwards by worker threads, each into its own node array (build_parallel).
    uint32_t defer_below = 0;
    std::vector<BuildTask> tasks;

    int build(uint32_t first, uint32_t count, uint32_t depth)
    {
        int me = (int)nodes.size();
        nodes.emplace_back();
        if (defer_below && depth > 0 && count <= defer_below) { tasks.push_back({ me, first, count, depth }); return me; }
        const bool big = count >= 65536u; // worth a parallel pass
        Box bb, cb;
        bb.reset(); cb.reset();
        if (big) {
            const size_t nc = n_chunks_of(count, 16384);
            std::vector<Box> pb(nc), pc(nc);
            parallel_chunks(count, 16384, [&](size_t b0, size_t e0, size_t c) {
                Box x, y; x.reset(); y.reset();
                for (size_t i = first + b0; i < first + e0; i++) { x.grow(tbox[order[i]]); y.grow(&cent[3 * (size_t)order[i]]); }
                pb[c] = x; pc[c] = y;
            });
            for (size_t c = 0; c < nc; c++) { bb.grow(pb[c]); cb.grow(pc[c]); }
        } else
            for (uint32_t i = first; i < first + count; i++) { bb.grow(tbox[order[i]]); cb.grow(&cent[3 * (size_t)order[i]]); }
        nodes[me].box = bb;
        max_depth = std::max(max_depth, depth);
        if (count <= leaf_max && (count <= 1 || depth > 0)) { // the root is always split so that it is an inner node
            nodes[me].first = first; nodes[me].count = count;
            return me;
        }
        // depth budget: with `levels` levels left a balanced tree must still fit
        uint32_t need = 0;
        while ((leaf_max << need) < count) need++;
        bool force_median = depth + need + 1 >= HJR_STACK_DEPTH;
        uint32_t mid = first;
        bool have = false;
        if (!force_median) {
            const int NB = 16;
            float best = FLT_MAX; int bax = -1, bsp = -1;
            struct Bins { Box box[16]; uint32_t cnt[16]; };
            for (int ax = 0; ax < 3; ax++) {
                float c0 = cb.lo[ax], c1 = cb.hi[ax];
                if (!(c1 > c0)) continue;
                Bins bins;
                for (int b = 0; b < NB; b++) { bins.box[b].reset(); bins.cnt[b] = 0; }
                float scale = (float)NB / (c1 - c0);
                auto fill = [&](Bins& B, size_t i0, size_t i1) {
                    for (size_t i = i0; i < i1; i++) {
                        int b = (int)((cent[3 * (size_t)order[i] + ax] - c0) * scale);
                        b = b < 0 ? 0 : (b >= NB ? NB - 1 : b);
                        B.box[b].grow(tbox[order[i]]); B.cnt[b]++;
                    }
                };
                if (big) {
                    const size_t nc = n_chunks_of(count, 16384);
                    std::vector<Bins> part(nc);
                    parallel_chunks(count, 16384, [&](size_t b0, size_t e0, size_t c) {
                        Bins& P = part[c];
                        for (int b = 0; b < NB; b++) { P.box[b].reset(); P.cnt[b] = 0; }
                        fill(P, first + b0, first + e0);
                    });
                    for (size_t c = 0; c < nc; c++)
                        for (int b = 0; b < NB; b++) { bins.box[b].grow(part[c].box[b]); bins.cnt[b] += part[c].cnt[b]; }
                } else fill(bins, first, (size_t)first + count);
                float rarea[NB]; uint32_t rcnt[NB];
                Box acc; acc.reset(); uint32_t n = 0;
                for (int b = NB - 1; b > 0; b--) { acc.grow(bins.box[b]); n += bins.cnt[b]; rarea[b] = acc.area(); rcnt[b] = n; }
                acc.reset(); n = 0;
                for (int b = 0; b < NB - 1; b++) {
                    acc.grow(bins.box[b]); n += bins.cnt[b];
                    if (n == 0 || rcnt[b + 1] == 0) continue;
                    float cost = acc.area() * (float)n + rarea[b + 1] * (float)rcnt[b + 1];
                    if (cost < best) { best = cost; bax = ax; bsp = b; }
                }
            }
            if (bax >= 0) {
                float c0 = cb.lo[bax], c1 = cb.hi[bax];
                float scale = 16.0f / (c1 - c0);
                uint32_t* it = std::partition(order + first, order + first + count, [&](uint32_t t) {
                    int b = (int)((cent[3 * (size_t)t + bax] - c0) * scale);
                    b = b < 0 ? 0 : (b >= 16 ? 15 : b);
                    return b <= bsp;
                });
                mid = (uint32_t)(it - order);
                have = mid > first && mid < first + count;
            }
        }
        if (!have) { // median split along the widest centroid axis (also the depth-budget fallback)
            int ax = 0;
            float ext = cb.hi[0] - cb.lo[0];
            for (int a = 1; a < 3; a++) if (cb.hi[a] - cb.lo[a] > ext) { ext = cb.hi[a] - cb.lo[a]; ax = a; }
            mid = first + count / 2;
            std::nth_element(order + first, order + mid, order + first + count,
                             [&](uint32_t a, uint32_t b) { return cent[3 * (size_t)a + ax] < cent[3 * (size_t)b + ax] || (cent[3 * (size_t)a + ax] == cent[3 * (size_t)b + ax] && a < b); });
        }
        int l = build(first, mid - first, depth + 1);
        int r = build(mid, first + count - mid, depth + 1);
        nodes[me].left = l; nodes[me].right = r;
        return me;
    }

    // Whole tree.  Scenes below `par_min` triangles: the plain recursion.  Above: the top of the tree is built by this thread
    // (its big nodes with parallel passes), the subtrees by worker threads; sub-results are spliced in task order, so the node
    // array (and everything emitted from it) is the same for any number of threads.
    bool timing = false; // stage times on stderr
    void build_all(uint32_t n, uint32_t par_min = 65536u, uint32_t task_size = 32768u)
    {
        nodes.reserve((size_t)2 * n);
        if (n < par_min) { build(0, n, 0); return; }
        auto t0 = std::chrono::steady_clock::now();
        defer_below = task_size;
        build(0, n, 0);
        defer_below = 0;
        if (timing) fprintf(stderr, "[hjr build]   top of the tree: %.1f ms, %zu subtree tasks, %u threads\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), tasks.size(), host_threads());
        std::vector<Builder> sub(tasks.size());
        parallel_chunks(tasks.size(), 1, [&](size_t b0, size_t e0, size_t) {
            for (size_t k = b0; k < e0; k++) {
                Builder& S = sub[k];
                S.leaf_max = leaf_max; S.tbox = tbox; S.cent = cent; S.order = order;
                S.nodes.reserve((size_t)2 * tasks[k].count);
                S.build(tasks[k].first, tasks[k].count, tasks[k].depth); // depth > 0: the sub-root may be a leaf
            }
        });
        if (timing) fprintf(stderr, "[hjr build]   + subtrees: %.1f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        // splice in task order: the sub-root replaces its placeholder, the other sub-nodes are appended (local i > 0 -> base + i)
        std::vector<size_t> base(tasks.size());
        size_t total = nodes.size();
        for (size_t k = 0; k < tasks.size(); k++) { base[k] = total - 1; total += sub[k].nodes.size() - 1; max_depth = std::max(max_depth, sub[k].max_depth); }
        nodes.resize(total);
        parallel_chunks(tasks.size(), 1, [&](size_t b0, size_t e0, size_t) {
            for (size_t k = b0; k < e0; k++) {
                const Builder& S = sub[k];
                auto remap = [&](int i) { return i < 0 ? i : (i == 0 ? tasks[k].node : (int)(base[k] + (size_t)i)); };
                for (size_t i = 0; i < S.nodes.size(); i++) {
                    BuildNode nd = S.nodes[i];
                    nd.left = remap(nd.left); nd.right = remap(nd.right);
                    nodes[i == 0 ? (size_t)tasks[k].node : base[k] + i] = nd;
                }
            }
        });
        tasks.clear();
        if (timing) fprintf(stderr, "[hjr build]   + splice: %.1f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
};

// Insertion-based refinement of a finished BVH2 (after Bittner, Hapala, Havran: "Fast insertion-based optimization of bounding volume
// hierarchies", CGF 2013): take a subtree out (its parent goes with it, the sibling moves up), find the place where putting it back adds
// the least surface area to the tree — branch and bound over (area added to the ancestors so far + area of the new common box) — and
// re-link it there with the freed parent node.  Subtrees are visited from the largest box down, for a few passes.  The cost model is
// the one the traversal kernels pay: one node step (both child boxes) per inner node entered, i.e. the sum of the inner nodes' areas;
// leaves (triangle ranges of `order`) are not touched, so triangle order and leaf records stay what build() made them.
// Hits do not depend on the tree (closest t, ties by primitive id): frames are the same bits with or without this pass.
#ifndef HJR_REFINE_PASSES
#define HJR_REFINE_PASSES 0     /* default passes, scenes up to 65536 triangles: none.  Bundled scene (984 triangles, 3 passes: 4 ms): 3.9 % less inner-node
                                   area, 1.9 % fewer node steps, NEE 112.95 -> 112.68 ms, but the tree gets deeper (12 -> 15 levels) and the wavefront
                                   kernel's stacks no longer fit LDS whole: MIS 172.0 -> 176.3 ms */
#endif
#ifndef HJR_REFINE_PASSES_BIG
#define HJR_REFINE_PASSES_BIG 1 /* above: one pass (1 M triangles: 105 ms next to a 250 ms build, 15.1 % less inner-node area, 17 % fewer node steps, NEE
                                   146.9 -> 134.7 ms, MIS 412 -> 339; three passes: 191 ms, 17.7 %, 134.3 ms) */
#endif
#ifndef HJR_REFINE_SHARE
#define HJR_REFINE_SHARE 0.01 /* share of the subtrees (largest boxes first) a pass tries to re-insert ... */
#endif
#ifndef HJR_REFINE_MIN_TAKE
#define HJR_REFINE_MIN_TAKE 2048u /* ... but at least this many: small scenes get full passes */
#endif
struct Refine {
    std::vector<BuildNode>& n;
    std::vector<int> parent;
    explicit Refine(std::vector<BuildNode>& nodes) : n(nodes), parent(nodes.size(), -1)
    {
        for (size_t i = 0; i < n.size(); i++)
            if (n[i].left >= 0) { parent[(size_t)n[i].left] = (int)i; parent[(size_t)n[i].right] = (int)i; }
    }
    static Box join(const Box& a, const Box& b) { Box r = a; r.grow(b); return r; }
    double inner_area() const
    {
        double s = 0.0;
        for (const BuildNode& x : n) if (x.left >= 0) s += (double)x.box.area();
        return s;
    }
    void refit_from(int i)
    {
        for (; i >= 0; i = parent[(size_t)i]) n[(size_t)i].box = join(n[(size_t)n[(size_t)i].left].box, n[(size_t)n[(size_t)i].right].box);
    }
    uint32_t depth_below(int root) const
    {
        uint32_t best = 0;
        std::vector<std::pair<int, uint32_t>> st{ { root, 0u } };
        while (!st.empty()) {
            auto [i, d] = st.back(); st.pop_back();
            best = std::max(best, d);
            if (n[(size_t)i].left >= 0) { st.push_back({ n[(size_t)i].left, d + 1 }); st.push_back({ n[(size_t)i].right, d + 1 }); }
        }
        return best;
    }
    // one pass over the `take` largest subtrees (by box area) whose parent is not the root; returns the number of subtrees that moved.
    // The large boxes near the top are where misplaced subtrees cost area: on the 1 M-triangle stress scene the largest 1 % of the nodes
    // bring 15.1 of the 16.1 % that a pass over all of them brings, in a tenth of the time.
    size_t pass(size_t take)
    {
        std::vector<std::pair<float, int>> byarea;
        byarea.reserve(n.size());
        for (size_t i = 1; i < n.size(); i++) if (parent[i] > 0) byarea.push_back({ n[i].box.area(), (int)i });
        auto larger = [](const std::pair<float, int>& a, const std::pair<float, int>& b) { return a.first > b.first || (a.first == b.first && a.second < b.second); };
        if (take < byarea.size()) { std::nth_element(byarea.begin(), byarea.begin() + (ptrdiff_t)take, byarea.end(), larger); byarea.resize(take); }
        std::sort(byarea.begin(), byarea.end(), larger);
        std::vector<int> cand(byarea.size());
        for (size_t i = 0; i < byarea.size(); i++) cand[i] = byarea[i].second;
        size_t moved = 0;
        std::vector<std::pair<float, int>> heap; // (area added to the ancestors, node): smallest first
        auto cmp = [](const std::pair<float, int>& a, const std::pair<float, int>& b) { return a.first > b.first || (a.first == b.first && a.second > b.second); };
        for (int L : cand) {
            const int P = parent[(size_t)L];
            if (P <= 0) continue; // an earlier move made its parent the root's child... or the root: leave it
            const int G = parent[(size_t)P];
            const int S = n[(size_t)P].left == L ? n[(size_t)P].right : n[(size_t)P].left;
            // detach: S takes P's place under G
            (n[(size_t)G].left == P ? n[(size_t)G].left : n[(size_t)G].right) = S;
            parent[(size_t)S] = G;
            refit_from(G);
            const Box lb = n[(size_t)L].box;
            const float la = lb.area();
            float best = FLT_MAX; int where = -1;
            heap.clear();
            heap.push_back({ 0.0f, 0 });
            while (!heap.empty()) {
                std::pop_heap(heap.begin(), heap.end(), cmp);
                const auto [inh, X] = heap.back(); heap.pop_back();
                if (inh + la >= best) break; // nothing below can beat the best place found
                const float direct = join(n[(size_t)X].box, lb).area();
                if (X != 0 && inh + direct < best) { best = inh + direct; where = X; } // (node 0 stays the root: nothing is put above it)
                if (n[(size_t)X].left >= 0) {
                    const float down = inh + (direct - n[(size_t)X].box.area());
                    if (down + la < best) {
                        heap.push_back({ down, n[(size_t)X].left }); std::push_heap(heap.begin(), heap.end(), cmp);
                        heap.push_back({ down, n[(size_t)X].right }); std::push_heap(heap.begin(), heap.end(), cmp);
                    }
                }
            }
            // re-link: P becomes the parent of (where, L) in where's old place
            const int X = where < 0 ? S : where;
            const int XP = parent[(size_t)X];
            n[(size_t)P].left = X; n[(size_t)P].right = L;
            parent[(size_t)X] = P; parent[(size_t)L] = P;
            parent[(size_t)P] = XP;
            if (XP >= 0) (n[(size_t)XP].left == X ? n[(size_t)XP].left : n[(size_t)XP].right) = P;
            refit_from(P);
            if (X != S) moved++;
        }
        return moved;
    }
};

} // namespace

namespace {

inline uint32_t leaf_ref(uint32_t first, uint32_t count) { return HJR_LEAF_FLAG | (count << 27) | first; }

// BVH2: inner nodes depth-first; a root that is itself a leaf (one triangle) gets an empty sibling
void emit_bvh2(const Builder& B, FrameData& out)
{
    std::vector<int> inner_id(B.nodes.size(), -1);
    uint32_t n_inner = 0;
    for (size_t i = 0; i < B.nodes.size(); i++)
        if (B.nodes[i].left >= 0) inner_id[i] = (int)n_inner++;
    if (n_inner == 0) {
        out.nodes.assign((size_t)HJR_NODE2_F4 * 4, 0.0f);
        const BuildNode& r = B.nodes[0];
        float* q = out.nodes.data();
        for (int c = 0; c < 2; c++)
            for (int a = 0; a < 3; a++) { q[4 * a + c] = r.box.lo[a]; q[4 * a + 2 + c] = r.box.hi[a]; }
        q[12] = q[14] = u2f(leaf_ref(r.first, r.count));
        q[13] = q[15] = u2f(leaf_ref(0, 0));
        out.n_nodes = 1;
        out.stack_need = 2;
        return;
    }
    out.nodes.assign((size_t)n_inner * HJR_NODE2_F4 * 4, 0.0f);
    out.n_nodes = n_inner;
    out.stack_need = B.max_depth + 2;
    for (size_t i = 0; i < B.nodes.size(); i++) {
        if (inner_id[i] < 0) continue;
        float* q = &out.nodes[(size_t)inner_id[i] * HJR_NODE2_F4 * 4];
        const int cid[2] = { B.nodes[i].left, B.nodes[i].right };
        for (int c = 0; c < 2; c++) {
            const BuildNode& ch = B.nodes[(size_t)cid[c]];
            for (int a = 0; a < 3; a++) { q[4 * a + c] = ch.box.lo[a]; q[4 * a + 2 + c] = ch.box.hi[a]; }
            q[12 + c] = q[14 + c] = u2f(ch.left >= 0 ? (uint32_t)inner_id[(size_t)cid[c]] : leaf_ref(ch.first, ch.count)); // stored twice (hjr_layout.h)
        }
    }
}

// BVH4: a wide node starts from the two children of a BVH2 inner node and repeatedly replaces its largest-area inner child by
// that child's two children until it has four children (or only leaves).  Breadth-first ids: the top of the tree is contiguous.
void emit_bvh4(const Builder& B, FrameData& out)
{
    struct Wide { int child[4]; int n; };
    std::vector<Wide> wide;
    std::vector<int> wide_of(B.nodes.size(), -1); // BVH2 inner node -> wide node id
    auto make_wide = [&](int root2) {
        Wide w;
        w.n = 0;
        if (B.nodes[(size_t)root2].left < 0) w.child[w.n++] = root2; // a single leaf: wide root with one leaf child
        else { w.child[w.n++] = B.nodes[(size_t)root2].left; w.child[w.n++] = B.nodes[(size_t)root2].right; }
        while (w.n < 4) {
            int best = -1;
            float barea = -1.0f;
            for (int c = 0; c < w.n; c++) {
                const BuildNode& ch = B.nodes[(size_t)w.child[c]];
                if (ch.left >= 0 && ch.box.area() > barea) { barea = ch.box.area(); best = c; }
            }
            if (best < 0) break;
            const int e = w.child[best];
            w.child[best] = B.nodes[(size_t)e].left;
            w.child[w.n++] = B.nodes[(size_t)e].right;
        }
        wide_of[(size_t)root2] = (int)wide.size();
        wide.push_back(w);
    };
    make_wide(0);
    for (size_t head = 0; head < wide.size(); head++)
        for (int c = 0; c < wide[head].n; c++)
            if (B.nodes[(size_t)wide[head].child[c]].left >= 0) make_wide(wide[head].child[c]);
    out.n_nodes = (uint32_t)wide.size();
    out.nodes.assign(wide.size() * (size_t)HJR_NODE4_F4 * 4, 0.0f);
    parallel_chunks(wide.size(), 8192, [&](size_t ib, size_t ie, size_t) {
    for (size_t i = ib; i < ie; i++) {
        float* q = &out.nodes[i * (size_t)HJR_NODE4_F4 * 4];
        for (int c = 0; c < 4; c++) {
            if (c < wide[i].n) {
                const BuildNode& ch = B.nodes[(size_t)wide[i].child[c]];
                for (int a = 0; a < 3; a++) { q[8 * a + c] = ch.box.lo[a]; q[8 * a + 4 + c] = ch.box.hi[a]; }
                q[24 + c] = u2f(ch.left >= 0 ? (uint32_t)wide_of[(size_t)wide[i].child[c]] : leaf_ref(ch.first, ch.count));
            } else { // unused slot: inverted box, empty leaf
                for (int a = 0; a < 3; a++) { q[8 * a + c] = 1e30f; q[8 * a + 4 + c] = -1e30f; }
                q[24 + c] = u2f(leaf_ref(0, 0));
            }
        }
    }
    });
    // exact worst-case traversal stack: every visited wide node can leave (children - 1) entries pending
    std::vector<uint32_t> pend(wide.size(), 0);
    uint32_t worst = 1;
    for (size_t i = 0; i < wide.size(); i++) { // parents precede children (breadth-first ids)
        const uint32_t here = pend[i] + (uint32_t)(wide[i].n > 0 ? wide[i].n - 1 : 0);
        worst = std::max(worst, here);
        for (int c = 0; c < wide[i].n; c++) {
            const int ch = wide[i].child[c];
            if (B.nodes[(size_t)ch].left >= 0) pend[(size_t)wide_of[(size_t)ch]] = here;
        }
    }
    out.stack_need = worst + 1;
}

} // namespace

void set_host_threads(int n) { g_host_threads.store(n > 0 ? n : 0, std::memory_order_relaxed); }

bool build_frame(const SceneCopy& sc, const float* M, const float* Mi, uint32_t n_inst, const BuildOptions& bo, FrameData& out, std::string& err)
{
    const bool allow_lds = bo.allow_lds;
    if (n_inst != sc.n_instances) { err = "instance count does not match the uploaded scene"; return false; }
    const uint32_t n = sc.n_triangles;
    const bool timing = bo.timing; // stage times on stderr
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[hjr build] %-22s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    out = FrameData();
    out.n_tris = n;
    out.tri_shade.assign((size_t)n * HJR_SHADE_F4 * 4, 0.0f);
    out.tri_inst.assign(n, 0);
    std::vector<float> wv((size_t)n * 9);

    // __closesthit__ch's per-hit work, done once per triangle (stale ptx:900-1263)
    parallel_chunks(n, 16384, [&](size_t tb, size_t te, size_t) {
        // instance of triangle tb: last prim_offset <= tb (empty instances share an offset with their successor)
        uint32_t i = (uint32_t)(std::upper_bound(sc.prim_offset.begin(), sc.prim_offset.begin() + n_inst, (uint32_t)tb) - sc.prim_offset.begin()) - 1;
        for (size_t t = tb; t < te; t++) {
            while (i + 1 < n_inst && sc.prim_offset[i + 1] <= t) i++;
            const float* m = M + 12 * (size_t)i;
            const float* mi = Mi + 12 * (size_t)i;
            float* s = &out.tri_shade[t * 16];
            float uv[6];
            for (int k = 0; k < 3; k++) {
                uint32_t idx = sc.indices[3 * t + k];
                V3 v = transform_position(m, { sc.vertices[3 * idx], sc.vertices[3 * idx + 1], sc.vertices[3 * idx + 2] });
                V3 nn = normalize(transform_normal(mi, { sc.normals[3 * idx], sc.normals[3 * idx + 1], sc.normals[3 * idx + 2] }));
                wv[9 * t + 3 * k + 0] = v.x; wv[9 * t + 3 * k + 1] = v.y; wv[9 * t + 3 * k + 2] = v.z;
                s[4 * k + 0] = nn.x; s[4 * k + 1] = nn.y; s[4 * k + 2] = nn.z;
                uv[2 * k] = sc.texcoords[2 * idx]; uv[2 * k + 1] = sc.texcoords[2 * idx + 1];
            }
            s[3] = uv[0]; s[7] = uv[1]; s[11] = uv[2];
            s[12] = uv[3]; s[13] = uv[4]; s[14] = uv[5];
            s[15] = u2f(sc.material_ids[t]);
            out.tri_inst[t] = i;
        }
    });

    lap("flatten");
    // light table (light_sample.h:22-58, 69-72)
    const uint32_t nl = (uint32_t)sc.light_prim_ids.size();
    out.n_lights = nl;
    out.lights.assign((size_t)nl * HJR_LIGHT_F4 * 4, 0.0f);
    for (uint32_t l = 0; l < nl; l++) {
        uint32_t prim = sc.light_prim_ids[l];
        // the reference's binary search over prim_offsets (light_sample.h:26-38) == last offset <= prim
        uint32_t inst = (uint32_t)(std::upper_bound(sc.prim_offset.begin(), sc.prim_offset.end(), prim) - sc.prim_offset.begin()) - 1;
        const float* m = M + 12 * inst;
        const float* mi = Mi + 12 * inst;
        V3 v[3], nn[3];
        for (int k = 0; k < 3; k++) {
            uint32_t idx = sc.indices[3 * prim + k];
            v[k] = transform_position(m, { sc.vertices[3 * idx], sc.vertices[3 * idx + 1], sc.vertices[3 * idx + 2] });
            nn[k] = transform_normal(mi, { sc.normals[3 * idx], sc.normals[3 * idx + 1], sc.normals[3 * idx + 2] });
        }
        V3 c = cross(sub(v[1], v[0]), sub(v[2], v[0]));
        float area = sqrtf(dot(c, c)) * 0.5f;
        float select_pdf = 1.0f / nl;
        float pdf = (float)(1.0 / (double)area); // `pdf = 1.0 / light_area;` is a double division (light_sample.h:69)
        pdf *= select_pdf;
        float* L = &out.lights[(size_t)l * 24];
        for (int k = 0; k < 3; k++) {
            L[4 * k + 0] = v[k].x; L[4 * k + 1] = v[k].y; L[4 * k + 2] = v[k].z;
            L[12 + 4 * k + 0] = nn[k].x; L[12 + 4 * k + 1] = nn[k].y; L[12 + 4 * k + 2] = nn[k].z;
        }
        L[3] = pdf;
        L[23] = 1.0f / pdf; // l5.w: float3 / pdf is float3 * (1.0f / pdf) (vec_math.h), the same IEEE division here as on the device
        L[19] = u2f(prim); // l4.w: global prim id (MIS looks the emissive triangle up by it)
        L[7] = sc.light_prim_emission[3 * l]; L[11] = sc.light_prim_emission[3 * l + 1]; L[15] = sc.light_prim_emission[3 * l + 2];
    }

    // BVH over padded triangle boxes
    Builder B;
    if (bo.leaf_max >= 1 && bo.leaf_max <= (int)HJR_LEAF_MAX) B.leaf_max = (uint32_t)bo.leaf_max; // option "leaf_max"
    B.timing = timing;
    std::vector<Box> tbox(n);
    std::vector<float> cent((size_t)n * 3);
    std::vector<uint32_t> order(n);
    float smax = 0.0f;
    {
        std::vector<float> part(n_chunks_of(n, 16384), 0.0f);
        parallel_chunks(n, 16384, [&](size_t tb, size_t te, size_t c) {
            float mx = 0.0f;
            for (size_t t = tb; t < te; t++) {
                Box b; b.reset();
                for (int k = 0; k < 3; k++) b.grow(&wv[9 * t + 3 * k]);
                for (int a = 0; a < 3; a++) {
                    cent[3 * t + a] = 0.5f * (b.lo[a] + b.hi[a]);
                    mx = std::max(mx, std::max(fabsf(b.lo[a]), fabsf(b.hi[a])));
                }
                tbox[t] = b; order[t] = (uint32_t)t;
            }
            part[c] = mx;
        });
        for (float v : part) { if (!(v <= smax)) smax = v; } // a NaN partial maximum must surface too
    }
    if (!(smax < 1e30f)) { err = "non-finite vertex after transform"; return false; }
    // conservative padding: the slab test must never cull a triangle the canonical ray/triangle test accepts
    float pad = smax * (1.0f / 32768.0f);
    parallel_chunks(n, 65536, [&](size_t tb, size_t te, size_t) {
        for (size_t t = tb; t < te; t++)
            for (int a = 0; a < 3; a++) { tbox[t].lo[a] -= pad; tbox[t].hi[a] += pad; }
    });
    B.tbox = tbox.data(); B.cent = cent.data(); B.order = order.data();
    lap("boxes");

    out.tri_geom.assign((size_t)std::max(n, 1u) * HJR_TRI_F4 * 4, 0.0f);
    if (n == 0) { // empty scene: one BVH4 root with four empty slots
        out.width = 4; out.lds_mode = 0;
        out.nodes.assign((size_t)HJR_NODE4_F4 * 4, 0.0f);
        for (int c = 0; c < 4; c++) {
            for (int a = 0; a < 3; a++) { out.nodes[8 * a + c] = 1e30f; out.nodes[8 * a + 4 + c] = -1e30f; }
            out.nodes[24 + c] = u2f(leaf_ref(0, 0));
        }
        out.n_nodes = 1;
        out.stack_need = 2;
        return true;
    }
    B.build_all(n);
    lap("bvh2 build");
    if (B.max_depth >= HJR_STACK_DEPTH) { err = "BVH deeper than the traversal stack"; return false; }
    { // option "bvh_refine": insertion-based refinement passes over the largest HJR_REFINE_SHARE of the subtrees (at least HJR_REFINE_MIN_TAKE)
        const int passes = bo.refine >= 0 ? bo.refine : (n <= 65536u ? HJR_REFINE_PASSES : HJR_REFINE_PASSES_BIG);
        if (passes > 0 && B.nodes.size() > 3) {
            // A tree that may be staged in LDS must fit the per-lane LDS stack columns (HJR_STACK_DEPTH levels): small scenes keep a copy to
            // fall back on.  A big scene is read from memory with short LDS stacks that overflow to HBM: any depth works there (emit_bvh4
            // sizes the overflow buffer exactly), and no copy is made.
            const bool small = n <= 65536u;
            std::vector<BuildNode> before;
            if (small) before = B.nodes;
            Refine R(B.nodes);
            const double a0 = timing || small ? R.inner_area() : 0.0;
            const size_t take = std::max<size_t>(HJR_REFINE_MIN_TAKE, (size_t)((double)B.nodes.size() * HJR_REFINE_SHARE));
            size_t moved = 0;
            for (int k = 0; k < passes; k++) { const size_t m = R.pass(take); moved += m; if (m == 0) break; }
            const uint32_t depth = R.depth_below(0);
            const double a1 = timing || small ? R.inner_area() : 0.0;
            const bool discard = small && (depth >= HJR_STACK_DEPTH || !(a1 < a0));
            if (discard) B.nodes = before; // too deep for the LDS stacks, or no gain: keep the built tree
            else B.max_depth = depth;
            if (timing) fprintf(stderr, "[hjr build]   refine: %zu subtrees moved, inner-node area %.4g -> %.4g (%.1f %%), depth %u%s\n", moved, a0, a1, 100.0 * (a1 / a0 - 1.0), depth, discard ? " (discarded)" : "");
            lap("bvh2 refine");
        }
    }
    out.depth = B.max_depth;
    parallel_chunks(n, 65536, [&](size_t kb, size_t ke, size_t) {
        for (size_t k = kb; k < ke; k++) {
            uint32_t t = order[k];
            float* g = &out.tri_geom[k * 12];
            memcpy(g, &wv[9 * (size_t)t], 9 * sizeof(float));
            g[9] = u2f(t);
            g[10] = u2f(sc.material_ids[t]); // copy of s3.w: the material fetch does not wait for the shading record
        }
    });
    lap("triangle reorder");
    // node format: BVH2 if tree + triangles + stacks fit into the LDS of one HJR_BLOCK_LDS-thread workgroup, else BVH4
    uint32_t n_inner2 = 0;
    for (size_t i = 0; i < B.nodes.size(); i++) if (B.nodes[i].left >= 0) n_inner2++;
    if (n_inner2 == 0) n_inner2 = 1;
    const size_t bvh2_bytes = (size_t)n_inner2 * HJR_NODE2_F4 * 16 + (size_t)n * HJR_TRI_F4 * 16;
    // the LDS variant also stages the material and light tables (hjr_kernel.hip.h): they count against the same budget
    const size_t table_bytes = sc.materials.size() * HJR_MAT_F4 * 16 + (size_t)out.n_lights * HJR_LIGHT_F4 * 16;
    const size_t stack2 = (size_t)B.max_depth + 2;
    int lds_mode = 0;
    if (allow_lds) {
        // 16-bit stack entries hold leaves of at most 3 triangles below triangle 8192 and node ids below 32768 (hjr_traverse.hip.h)
        const bool fits16 = (size_t)HJR_BLOCK_LDS * stack2 * 2 + 16 + bvh2_bytes + table_bytes <= HJR_LDS_BUDGET && n_inner2 < 32768u && n < 8192u && B.leaf_max <= 3u;
        const bool prefer16 = bo.prefer_stack16; // option "lds_stack16": 16-bit entries whenever they fit
        if (fits16 && prefer16) lds_mode = 2;
        else if ((size_t)HJR_BLOCK_LDS * stack2 * 4 + 16 + bvh2_bytes + table_bytes <= HJR_LDS_BUDGET) lds_mode = 1;
        else if (fits16) lds_mode = 2;
    }
    if (bo.bvh_width == 2 || bo.bvh_width == 4) { // option "bvh_width": force a node format (forcing 4 also forces the memory path)
        const int v = bo.bvh_width;
        if (v == 4) lds_mode = 0;
        out.width = (v == 2 || (v != 4 && lds_mode)) ? 2u : 4u;
    } else out.width = lds_mode ? 2u : 4u;
    out.lds_mode = lds_mode;
    if (out.width == 2) emit_bvh2(B, out);
    else emit_bvh4(B, out);
    lap("emit nodes");
    return true;
}

} // namespace hjr
