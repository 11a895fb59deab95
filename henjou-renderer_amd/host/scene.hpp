// Host-side scene model: mirrors the reference's SceneData / Material / Animation / RenderOption
// (renderer/scene.h:9-36, renderer/material.h:10-63, renderer/animation.h:20-94, renderer/render_option.h:45-84)
// so that the loaders and the frame set-up read like the reference's.  All arithmetic is fp32 in the
// reference's evaluation order (this translation unit is compiled with -ffp-contract=off).
#pragma once
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

// common/matrix.h:6-19 — row-major 4x4
struct Affine4x4 {
    float v[16];
    Affine4x4() { for (float& f : v) f = 0; }
    float operator[](int i) const { return v[i]; }
};
inline Affine4x4 translateAffine(const float3_& t) // matrix.h:21-24
{
    Affine4x4 a;
    const float v[16] = { 1, 0, 0, t.x, 0, 1, 0, t.y, 0, 0, 1, t.z, 0, 0, 0, 1 };
    for (int i = 0; i < 16; i++) a.v[i] = v[i];
    return a;
}
inline Affine4x4 scaleAffine(const float3_& s) // matrix.h:26-29
{
    Affine4x4 a;
    const float v[16] = { s.x, 0, 0, 0, 0, s.y, 0, 0, 0, 0, s.z, 0, 0, 0, 0, 1 };
    for (int i = 0; i < 16; i++) a.v[i] = v[i];
    return a;
}
inline Affine4x4 rotateAffine(const float4_& q) // matrix.h:32-56 (the `2.0 *` products are evaluated in double there)
{
    float q2xy = (float)(2.0 * q.x * q.y);
    float q2xz = (float)(2.0 * q.x * q.z);
    float q2xw = (float)(2.0 * q.x * q.w);
    float q2yz = (float)(2.0 * q.y * q.z);
    float q2yw = (float)(2.0 * q.y * q.w);
    float q2zw = (float)(2.0 * q.z * q.w);
    float q2ww = (float)(2.0 * q.w * q.w);
    Affine4x4 a;
    const float v[16] = { q2ww + 2.0f * q.x * q.x - 1.0f, q2xy - q2zw, q2xz + q2yw, 0,
                          q2xy + q2zw, q2ww + 2.0f * q.y * q.y - 1.0f, q2yz - q2xw, 0,
                          q2xz - q2yw, q2yz + q2xw, q2ww + 2.0f * q.z * q.z - 1.0f, 0,
                          0, 0, 0, 1 };
    for (int i = 0; i < 16; i++) a.v[i] = v[i];
    return a;
}
inline float4_ operator*(const Affine4x4& a, const float4_& p) // matrix.h:58-65
{
    return { p.x * a[0] + p.y * a[1] + p.z * a[2] + p.w * a[3], p.x * a[4] + p.y * a[5] + p.z * a[6] + p.w * a[7],
             p.x * a[8] + p.y * a[9] + p.z * a[10] + p.w * a[11], p.x * a[12] + p.y * a[13] + p.z * a[14] + p.w * a[15] };
}
inline Affine4x4 operator*(const Affine4x4& a, const Affine4x4& b) // matrix.h:67-76
{
    Affine4x4 r;
    for (int j = 0; j < 4; j++)
        for (int i = 0; i < 4; i++)
            r.v[i + j * 4] = a[0 + j * 4] * b[i + 0 * 4] + a[1 + j * 4] * b[i + 1 * 4] + a[2 + j * 4] * b[i + 2 * 4] + a[3 + j * 4] * b[i + 3 * 4];
    return r;
}

// renderer/animation.h:20-32
template <typename T> struct AnimationData {
    std::vector<T> data;
    std::vector<float> key;
};

// renderer/animation.h:34-131.  Interpolation is LINEAR for every sampler (the glTF "interpolation" field is never
// read, gltfloader.h:1538-1588) and quaternions are lerped without re-normalisation (animation.h:70-79).
struct Animation {
    AnimationData<float3_> translation_data;
    AnimationData<float4_> rotation_data;
    AnimationData<float3_> scale_data;

    template <typename T> static T animationInterpolate(const std::vector<T>& animation, const std::vector<float>& key, float time)
    { // animation.h:42-67
        if (key.size() == 1 || time < 0) return animation[0];
        int first = 0, len = (int)key.size();
        while (len > 0) {
            int half = len >> 1, middle = first + half;
            if (key[middle] <= time) { first = middle + 1; len -= half + 1; }
            else len = half;
        }
        int offset = first - 1;
        if (key.size() - 1 <= (size_t)offset) return animation[key.size() - 1]; // sic: offset == -1 wraps and also lands here
        float time_offset = time - key[offset];
        float time_delta = key[offset + 1] - key[offset];
        float delta = time_offset / time_delta;
        return animation[offset] * (1.0f - delta) + animation[offset + 1] * (delta);
    }
    float3_ translation(float time) const
    {
        return translation_data.key.size() ? animationInterpolate(translation_data.data, translation_data.key, time) : float3_{ 0, 0, 0 };
    }
    float4_ rotation(float time) const
    {
        return rotation_data.key.size() ? animationInterpolate(rotation_data.data, rotation_data.key, time) : float4_{ 0, 0, 0, 0 };
    }
    float3_ scale(float time) const
    {
        return scale_data.key.size() ? animationInterpolate(scale_data.data, scale_data.key, time) : float3_{ 0, 0, 0 };
    }
    Affine4x4 getAnimationAffine(float time) const // animation.h:81-94: T * R * S, no node hierarchy
    {
        return translateAffine(translation(time)) * rotateAffine(rotation(time)) * scaleAffine(scale(time));
    }
    Affine4x4 getRotateAnimationAffine(float time) const { return rotateAffine(rotation(time)); } // animation.h:96-103
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
    std::vector<Animation> animations;
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
