// Host-side scene model: the data of the reference's SceneData / Material / RenderOption (renderer/scene.h:9-36,
// renderer/material.h:10-63, renderer/render_option.h:45-84) and the node motion that renderer/animation.h:20-131 and
// common/matrix.h:6-104 evaluate per frame.  All arithmetic is fp32 in the reference's evaluation order (every translation
// unit that includes this header is compiled with -ffp-contract=off).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/henjou_hip.h"

namespace hjr {

struct float2_ { float x, y; };
struct float3_ { float x, y, z; };
struct float4_ { float x, y, z, w; };

inline float3_ operator*(const float3_& a, float s) { return { a.x * s, a.y * s, a.z * s }; }
inline float3_ operator+(const float3_& a, const float3_& b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
inline float3_ operator-(const float3_& a, const float3_& b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
inline float4_ operator*(const float4_& a, float s) { return { a.x * s, a.y * s, a.z * s, a.w * s }; }
inline float4_ operator+(const float4_& a, const float4_& b) { return { a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w }; }
inline float dot3(const float3_& a, const float3_& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float3_ cross3(const float3_& a, const float3_& b)
{
    return { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x };
}
inline float3_ normalize3(const float3_& v) // sutil/vec_math.h normalize: v * (1/sqrt(dot))
{
    float invLen = 1.0f / sqrtf(dot3(v, v));
    return v * invLen;
}

// ---- node motion: keyframe tracks -> per-frame instance matrix.
// What must match the reference bit for bit is the VALUE of every matrix entry (common/matrix.h:21-76 builds T, R and S as
// 4x4 matrices and multiplies them; renderer/animation.h:42-103 samples the keys), not its data structures.  Here a node
// keeps three keyframe tracks and writes the 3x4 matrix the kernel-side layout wants directly; every entry is the same
// left-to-right sum of the same four products the 4x4 multiplications form (zero and one factors included, so signed zeros
// and non-finite inputs propagate identically).

// One animated channel of a node.  Key 0 is the node's static TRS at time 0; the glTF channels are appended behind it
// (gltfloader.h:1313-1343, 1536-1590).  Sampling is linear for every sampler (the glTF "interpolation" field is never read)
// and component-wise, also for quaternions, which are NOT re-normalised (animation.h:70-79).
template <typename V> struct Track {
    std::vector<float> times;
    std::vector<V> values;
    void push(float t, const V& v) { times.push_back(t); values.push_back(v); }
    bool empty() const { return times.empty(); }
    // animation.h:42-67.  `next` = number of keys <= time (the reference's hand-written bisection is exactly this
    // upper bound).  Quirk kept on purpose: with several keys and 0 <= time < times[0] the reference's `offset = next - 1`
    // is -1, wraps in its unsigned comparison and selects the LAST key, the same as a time past the end.
    V at(float time) const
    {
        if (times.size() == 1 || time < 0) return values[0];
        const size_t next = (size_t)(std::upper_bound(times.begin(), times.end(), time, [](float t, float key) { return !(key <= t); }) - times.begin());
        const bool before_first_key = next == 0, at_or_past_last_key = next >= times.size();
        if (before_first_key || at_or_past_last_key) return values[times.size() - 1];
        const size_t k = next - 1;
        const float w = (time - times[k]) / (times[k + 1] - times[k]);
        return values[k] * (1.0f - w) + values[k + 1] * w;
    }
};

struct NodeMotion {
    Track<float3_> translation;
    Track<float4_> rotation;
    Track<float3_> scale;

    // rotation block of matrix.h:32-56: the `2.0 * a * b` products are double expressions there, rounded to float once
    static void rotation_rows(const float4_& q, float r[3][3])
    {
        const float xy = (float)(2.0 * q.x * q.y), xz = (float)(2.0 * q.x * q.z), xw = (float)(2.0 * q.x * q.w);
        const float yz = (float)(2.0 * q.y * q.z), yw = (float)(2.0 * q.y * q.w), zw = (float)(2.0 * q.z * q.w);
        const float ww = (float)(2.0 * q.w * q.w);
        r[0][0] = ww + 2.0f * q.x * q.x - 1.0f; r[0][1] = xy - zw;                       r[0][2] = xz + yw;
        r[1][0] = xy + zw;                       r[1][1] = ww + 2.0f * q.y * q.y - 1.0f; r[1][2] = yz - xw;
        r[2][0] = xz - yw;                       r[2][1] = yz + xw;                       r[2][2] = ww + 2.0f * q.z * q.z - 1.0f;
    }
    // Rows 0..2 of (T * R) * S (animation.h:81-94: no node hierarchy), written as a row-major 3x4.
    void matrix3x4(float time, float out[12]) const
    {
        const float3_ t = translation.empty() ? float3_{ 0, 0, 0 } : translation.at(time);
        const float4_ q = rotation.empty() ? float4_{ 0, 0, 0, 0 } : rotation.at(time);
        const float3_ s = scale.empty() ? float3_{ 0, 0, 0 } : scale.at(time);
        float r[3][3];
        rotation_rows(q, r);
        const float tv[3] = { t.x, t.y, t.z }, sv[3] = { s.x, s.y, s.z };
        auto R = [&](int k, int i) { return (k < 3 && i < 3) ? r[k][i] : ((k == 3 && i == 3) ? 1.0f : 0.0f); }; // [R 0; 0 1]
        auto S = [&](int k, int i) { return k != i ? 0.0f : (k < 3 ? sv[k] : 1.0f); };                          // diag(s, 1)
        for (int j = 0; j < 3; j++) {
            float tr[4]; // row j of T * R; row j of T is (e_j, t_j)
            for (int i = 0; i < 4; i++) {
                float acc = (j == 0 ? 1.0f : 0.0f) * R(0, i);
                acc = acc + (j == 1 ? 1.0f : 0.0f) * R(1, i);
                acc = acc + (j == 2 ? 1.0f : 0.0f) * R(2, i);
                acc = acc + tv[j] * R(3, i);
                tr[i] = acc;
            }
            for (int i = 0; i < 4; i++) {
                float acc = tr[0] * S(0, i);
                acc = acc + tr[1] * S(1, i);
                acc = acc + tr[2] * S(2, i);
                acc = acc + tr[3] * S(3, i);
                out[4 * j + i] = acc;
            }
        }
    }
    // animation.h:96-103: the rotation alone (camera direction / up)
    void rotation3x3(float time, float r[3][3]) const { rotation_rows(rotation.empty() ? float4_{ 0, 0, 0, 0 } : rotation.at(time), r); }
};

struct GeometryData { uint32_t index_offset, index_count; };   // scene.h:9-12
struct InstanceData { uint32_t geometry_id, animation_id; };   // scene.h:14-17

struct SceneData { // scene.h:19-36
    std::vector<float3_> vertices;
    std::vector<uint32_t> indices;
    std::vector<uint32_t> material_ids;
    std::vector<float3_> normals;
    std::vector<float2_> texcoords;
    std::vector<hjr_material> materials;
    std::vector<std::string> material_names;
    std::vector<std::string> texture_files; // de-duplicated by name (texture_load.h:7-20)
    struct TexturePixels { std::vector<uint8_t> rgba; uint32_t width = 0, height = 0; int srgb = 1; };
    std::vector<TexturePixels> textures;    // Texture (renderer/texture.h:16-39), same slots as texture_files
    std::vector<hjr_texture> texture_views;
    std::vector<uint32_t> light_prim_ids;
    std::vector<float3_> light_prim_emission;
    std::vector<NodeMotion> animations; // one per glTF node (InstanceData.animation_id indexes it)
    std::vector<GeometryData> geometries;
    std::vector<InstanceData> instances;
    std::vector<uint32_t> prim_offset;
    // flattened per-instance columns for the C view
    std::vector<uint32_t> geo_index_offset, geo_index_count, inst_animation_id;
};

// 3x4 inverse of an affine matrix (replaces glm::inverse in renderer.h:274-283, whose exact rounding is not observable).
void affine_inverse_3x4(const float* m12, float* inv12);

bool load_render_option(const std::string& path, hjr_render_option& opt, std::string& err);
bool load_gltf(const std::string& dir, const std::string& file, SceneData& scene, hjr_render_option& opt, std::string& err);
void eval_transforms(const SceneData& scene, float time, float* m12, float* inv12);
void eval_camera(const SceneData& scene, const hjr_render_option& opt, float time, hjr_camera& cam);

} // namespace hjr
