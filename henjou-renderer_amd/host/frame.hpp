// Per-frame host set-up that replaces the reference's GAS/IAS build (renderer/renderer.h:319-490) and hoists the
// per-hit vertex/normal transforms of __closesthit__ch and light_sample (light_sample.h:43-58) to once per frame:
// instances are flattened to world space, a binned-SAH BVH2 is built over all triangles, and the light table is
// pre-transformed.  The arithmetic of every value the kernel consumes is the reference's (fp32, same order).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/henjou_hip.h"
#include "../csrc/hjr_layout.h"
#include "options.hpp"

namespace hjr {

struct FrameData {
    std::vector<float> nodes;      // HJR_NODE2_F4 / HJR_NODE4_F4 float4 per inner node
    std::vector<float> tri_geom;   // HJR_TRI_F4 float4 per triangle, leaf order
    std::vector<float> tri_shade;  // HJR_SHADE_F4 float4 per triangle, global prim order
    std::vector<uint32_t> tri_inst;// instance id per global prim
    std::vector<float> lights;     // HJR_LIGHT_F4 float4 per emissive triangle
    uint32_t n_tris = 0, n_nodes = 0, n_lights = 0, depth = 0;
    uint32_t stack_need = 2; // worst-case traversal stack entries per lane for this tree
    uint32_t width = 2;      // 2 or 4 (node format, hjr_layout.h)
    int lds_mode = 0;        // 0: nodes/triangles read from memory; 1: staged in LDS, 32-bit stack entries; 2: 16-bit entries
};

// Owning copy of an hjr_scene_view (cpySceneDataToDevice keeps the host vectors alive too, renderer.h:197-255).
struct SceneCopy {
    std::vector<float> vertices, normals, texcoords, light_prim_emission;
    std::vector<uint32_t> indices, material_ids, prim_offset, light_prim_ids;
    std::vector<hjr_material> materials;
    struct Tex { uint32_t offset, width, height; int srgb; }; // offset in texels into `texels`
    std::vector<Tex> textures;
    std::vector<uint32_t> texels; // RGBA8 atlas, all textures back to back
    uint32_t n_triangles = 0, n_instances = 0;
    bool set(const hjr_scene_view& v, std::string& err);
};

// bo.allow_lds: let small scenes use the LDS-resident BVH2 layout; the other fields force a layout / leaf size (tests, tuning)
bool build_frame(const SceneCopy& sc, const float* transforms12, const float* inv12, uint32_t n_instances, const BuildOptions& bo,
                 FrameData& out, std::string& err);

} // namespace hjr
