// The small kernels around the render kernels: cost-ordered tile list, chunk-sum finalisation, tile packing for the multi-GPU gather.
// Included by hjr_device.hip only (non-template __global__ functions: one translation unit).
#pragma once
#include "hjr_kernel.hip.h"

// ---- cost-ordered tile list.  The frame ends when the slowest work item ends, and an item (8 samples of one pixel, each up
// to 10 bounces, strictly sequential) can run for milliseconds: with plain scanline order the tail of the launch is whatever
// the last tiles happen to cost (5 ms on a 64 x 64 frame, 15 % of an 18 ms launch when the frame is split over 8 GPUs).
// One wave per owned tile casts the 64 pixel-centre rays (no RNG), classifies the tile by its costliest first hit
// (3 = glass, 2 = metallic, 1 = other surface, 0 = background or light) and the tiles are handed out class 3 first, background
// last: longest-processing-time-first scheduling, and waves whose lanes behave alike.  Only the ORDER of the work changes;
// every pixel is computed exactly as before.
template <int WIDTH>
__global__ void __launch_bounds__(64) hjr_classify_tiles_kernel(const KParams P)
{
    typedef LaneStack<uint32_t, 64, false> ST;
    ST stack;
    stack.lds = reinterpret_cast<uint32_t*>(hjr_smem) + threadIdx.x;
    stack.spill = nullptr; stack.spill_stride = 0; stack.lds_n = 0; stack.n_over = 0; stack.top = nullptr; stack.n_top = 0u;
    uint32_t n_cls[4] = { 0u, 0u, 0u, 0u }; // per block; one atomic per class at the end (32 k atomics on four words cost 0.4 ms)
    for (uint32_t idx = blockIdx.x; idx < P.n_owned_tiles; idx += gridDim.x) {
        const uint32_t tile = idx * P.world + P.rank;
        uint32_t tx, ty;
        hjr_tile_xy(tile, P.tiles_x, &tx, &ty);
        const uint32_t px = tx * HJR_TILE + (threadIdx.x & 7u), py = ty * HJR_TILE + (threadIdx.x >> 3);
        uint32_t cls = 0;
        if (px < P.width && py < P.height) {
            const float W = (float)P.width, H = (float)P.height;
            const float u = (2.0f * ((float)px + 0.5f) - W) / H, v = (2.0f * ((float)py + 0.5f) - H) / H;
            const f3 cd = V(P.cam_dir[0], P.cam_dir[1], P.cam_dir[2]), cu = V(P.cam_up[0], P.cam_up[1], P.cam_up[2]);
            const f3 cr = V(P.cam_right[0], P.cam_right[1], P.cam_right[2]);
            const f3 d = normalize(cd * P.cam_f + cr * u + cu * v);
            Hit h;
            Counters cnt; cnt.box = cnt.tri = 0;
            if (traverse<false, false, WIDTH, 64, ST>(P.nodes, P.tri_geom, V(P.cam_pos[0], P.cam_pos[1], P.cam_pos[2]), d, 0.001f, 1e16f, h, stack, cnt)) {
                const float4* m = P.materials + f2bits(P.tri_geom[h.k * HJR_TRI_F4 + 2].z) * HJR_MAT_F4;
                const float4 m0 = m[0], m3 = m[3];
                cls = f2bits(m3.x) != 0 ? 0u : (f2bits(m3.y) != 0 ? 3u : (m0.w > 0.5f ? 2u : 1u));
            }
        }
        const uint32_t tcls = __ballot(cls == 3u) ? 3u : (__ballot(cls == 2u) ? 2u : (__ballot(cls == 1u) ? 1u : 0u));
        if (threadIdx.x == 0) P.tile_class[idx] = tcls;
        n_cls[0] += tcls == 0u; n_cls[1] += tcls == 1u; n_cls[2] += tcls == 2u; n_cls[3] += tcls == 3u;
    }
    if (threadIdx.x < 4u && n_cls[threadIdx.x]) atomicAdd(&P.tile_count[threadIdx.x], n_cls[threadIdx.x]);
}
__global__ void __launch_bounds__(256) hjr_order_tiles_kernel(const KParams P)
{
    const uint32_t idx = blockIdx.x * 256u + threadIdx.x;
    const bool live = idx < P.n_owned_tiles;
    const uint32_t cls = live ? P.tile_class[idx] : 0xffffffffu;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t pos = 0;
    for (uint32_t c = 0; c < 4u; c++) { // wave-aggregated scatter: one atomic per wave and class
        const unsigned long long m = __ballot(cls == c);
        if (m == 0ull) continue;
        uint32_t base = 0;
        if (lane == (uint32_t)(__ffsll((long long)m) - 1)) base = atomicAdd(&P.tile_count[4 + c], (uint32_t)__popcll(m));
        base = (uint32_t)__shfl((int)base, __ffsll((long long)m) - 1);
        if (cls == c) {
            uint32_t first = 0;
            for (uint32_t k = 3u; k > c; k--) first += P.tile_count[k];
            pos = first + base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        }
    }
    if (live) P.tile_order_w[pos] = idx * P.world + P.rank;
}

// From the second frame of a sequence on, the tiles are ordered by what they actually cost in the previous frame (closest-hit
// rays per sample) inside their first-hit class: a counting sort over 64 keys in two kernels; the order inside a key is arbitrary.
HD uint32_t cost_bucket(uint32_t cls, uint32_t cost, uint32_t cost_div)
{
    // key = (first-hit class, measured rays per sample in steps of 1/2): the class keeps waves of like materials together in
    // time (3 % at N = 1), the cost orders the tiles inside a class so that the last items of a class are its cheapest
    const uint32_t b = (uint32_t)(((unsigned long long)cost * 2ull) / cost_div);
    return (cls & 3u) * 16u + (b > 15u ? 15u : b);
}
__global__ void __launch_bounds__(256) hjr_cost_hist_kernel(const KParams P)
{
    __shared__ uint32_t h[64];
    if (threadIdx.x < 64u) h[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t idx = blockIdx.x * 256u + threadIdx.x;
    if (idx < P.n_owned_tiles) {
        const uint32_t b = cost_bucket(P.tile_class[idx], P.tile_cost[idx], P.cost_div);
        P.tile_bucket[idx] = b;
        atomicAdd(&h[b], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 64u && h[threadIdx.x]) atomicAdd(&P.cost_hist[threadIdx.x], h[threadIdx.x]);
}
__global__ void __launch_bounds__(256) hjr_cost_scatter_kernel(const KParams P)
{
    __shared__ uint32_t h[64], base[64];
    if (threadIdx.x < 64u) h[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t idx = blockIdx.x * 256u + threadIdx.x;
    const bool live = idx < P.n_owned_tiles;
    uint32_t b = 0, rank_in_block = 0;
    if (live) { b = P.tile_bucket[idx]; rank_in_block = atomicAdd(&h[b], 1u); P.tile_cost[idx] = 0u; } // zeroed for this frame's sums
    __syncthreads();
    if (threadIdx.x < 64u) {
        uint32_t first = 0;
        for (uint32_t k = 63u; k > threadIdx.x; k--) first += P.cost_hist[k]; // expensive buckets first
        base[threadIdx.x] = h[threadIdx.x] ? first + atomicAdd(&P.cost_hist[64u + threadIdx.x], h[threadIdx.x]) : 0u;
    }
    __syncthreads();
    if (live) P.tile_order_w[base[b] + rank_in_block] = idx * P.world + P.rank;
}

// Adds the chunk sums of every owned pixel in chunk order and scales by 1/spp: a fixed summation
// tree, so the frame is bitwise independent of which lane/wave/GPU rendered which chunk.  Streaming kernel: one lane
// per pixel of an owned tile, n_chunks coalesced float4 loads ([chunk][owned tile][64] layout), one float4 store.
__global__ void __launch_bounds__(256) hjr_finalize_kernel(const KParams P)
{
    const size_t n_slots = (size_t)P.n_owned_tiles * 64u; // chunk-sum slots per chunk: 64 per owned tile
    const float inv_spp = 1.0f / (float)P.spp;
    for (size_t sl = (size_t)blockIdx.x * blockDim.x + threadIdx.x; sl < n_slots; sl += (size_t)gridDim.x * blockDim.x) {
        const uint32_t tile = (uint32_t)(sl >> 6) * P.world + P.rank;
        uint32_t tx, ty;
        hjr_tile_xy(tile, P.tiles_x, &tx, &ty);
        const uint32_t x = tx * HJR_TILE + ((uint32_t)sl & 7u), y = ty * HJR_TILE + (((uint32_t)sl >> 3) & 7u);
        if (x >= P.width || y >= P.height) continue;
        const size_t pix = P.packed ? sl : (size_t)y * P.width + x;
        float4 a = make_float4(0.0f, 0.0f, 0.0f, 0.0f), b = a, c = a;
        for (uint32_t k = 0; k < P.n_chunks; k++) {
            const float4 v = P.part_color[(size_t)k * n_slots + sl];
            a.x = a.x + v.x; a.y = a.y + v.y; a.z = a.z + v.z;
            if (P.aov_albedo) { const float4 w = P.part_albedo[(size_t)k * n_slots + sl]; b.x = b.x + w.x; b.y = b.y + w.y; b.z = b.z + w.z; }
            if (P.aov_normal) { const float4 w = P.part_normal[(size_t)k * n_slots + sl]; c.x = c.x + w.x; c.y = c.y + w.y; c.z = c.z + w.z; }
        }
        P.aov_color[pix] = make_float4(a.x * inv_spp, a.y * inv_spp, a.z * inv_spp, 1.0f);
        if (P.aov_albedo) P.aov_albedo[pix] = make_float4(b.x * inv_spp, b.y * inv_spp, b.z * inv_spp, 1.0f);
        if (P.aov_normal) P.aov_normal[pix] = make_float4(c.x * inv_spp, c.y * inv_spp, c.z * inv_spp, 1.0f);
    }
}

// ---- tile pack / unpack: the multi-GPU exchange moves only owned tiles ([owned tile][64] float4)
__global__ void __launch_bounds__(256) hjr_pack_tiles_kernel(const float4* frame, float4* packed, uint32_t width, uint32_t height, uint32_t tiles_x, uint32_t n_owned, uint32_t rank, uint32_t world)
{
    const size_t sl = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (sl >= (size_t)n_owned * 64u) return;
    const uint32_t tile = (uint32_t)(sl >> 6) * world + rank;
    uint32_t tx, ty;
    hjr_tile_xy(tile, tiles_x, &tx, &ty);
    const uint32_t x = tx * HJR_TILE + ((uint32_t)sl & 7u), y = ty * HJR_TILE + (((uint32_t)sl >> 3) & 7u);
    packed[sl] = (x < width && y < height) ? frame[(size_t)y * width + x] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}
__global__ void __launch_bounds__(256) hjr_unpack_tiles_kernel(const float4* packed, float4* frame, uint32_t width, uint32_t height, uint32_t tiles_x, uint32_t n_owned, uint32_t rank, uint32_t world)
{
    const size_t sl = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (sl >= (size_t)n_owned * 64u) return;
    const uint32_t tile = (uint32_t)(sl >> 6) * world + rank;
    uint32_t tx, ty;
    hjr_tile_xy(tile, tiles_x, &tx, &ty);
    const uint32_t x = tx * HJR_TILE + ((uint32_t)sl & 7u), y = ty * HJR_TILE + (((uint32_t)sl >> 3) & 7u);
    if (x < width && y < height) frame[(size_t)y * width + x] = packed[sl];
}

// ---- the 8-bit preview buffer of the raygen program (`uchar4* image` of Params, renderer/renderer.h:1102, 1175: written by the missing
// __raygen__rg, never read back by the host — the PNG comes from AOV_Output).  Build-defined: the colour AOV through the tonemappers of
// kernel/color.h (Tonemap_Uchimura :10-39, ACESFilm :55-63), then toSRGB + quantizeUnsignedChar as float4ConvertColor does on the host
// (renderer.h:73-101).  The host form is hjr_tonemap_to_srgb8 (libm); this one uses the device's pow / exp, so single pixels may differ by
// one code value at a quantisation boundary.
__device__ inline float hjr_tm_uchimura(float x)
{
    const float P = 1.0f, a = 1.0f, m = 0.22f, l = 0.4f, c = 1.33f, b = 0.0f;
    const float l0 = ((P - m) * l) / a, S0 = m + l0, S1 = m + a * l0, C2 = (a * P) / (P - S1), CP = -C2 / P;
    const float sx = fmaxf(0.0f, fminf((x - 0.0f) / (m - 0.0f), 1.0f));
    const float w0 = (float)(1.0 - (double)(sx * sx * (3.0f - 2.0f * sx)));
    const float w2 = (float)((m + l0) < x);
    const float w1 = (float)(1.0 - (double)w0 - (double)w2);
    const float T = (float)((double)m * pow((double)(x / m), (double)c) + (double)b);
    const float S = (float)((double)P - (double)(P - S1) * exp((double)(CP * (x - S0))));
    const float L = m + a * (x - m);
    return T * w0 + L * w1 + S * w2;
}
__device__ inline float hjr_tm_aces(float x)
{
    const float a = 2.51f, b = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;
    return fmaxf(0.0f, fminf((x * (a * x + b)) / (x * (c * x + d) + e), 1.0f));
}
__global__ void __launch_bounds__(256) hjr_preview_kernel(const float4* color, uchar4* out, size_t n, int tonemap)
{
    const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float4 v = color[i];
    float ch[3] = { v.x, v.y, v.z };
    unsigned char q8[3];
    for (int k = 0; k < 3; k++) {
        float col = ch[k];
        col = tonemap == 1 ? hjr_tm_uchimura(col) : (tonemap == 2 ? hjr_tm_aces(col) : col);
        const float powed = powf(col, 1.0f / 2.4f);
        const float sr = col < 0.0031308f ? 12.92f * col : 1.055f * powed - 0.055f;
        const float q = sr * 256.0f;
        const uint32_t u = (q > 0.0f) ? ((q >= 4294967040.0f) ? 4294967040u : (uint32_t)q) : 0u;
        q8[k] = (unsigned char)(u < 255u ? u : 255u);
    }
    out[i] = make_uchar4(q8[0], q8[1], q8[2], 255);
}

