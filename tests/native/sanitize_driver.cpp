// AddressSanitizer / UBSan run of the product's host code (scene surface, BVH builder, image io) and of the oracle on the
// CPU.  Built and executed by tests/test_sanitizers.py; GPU sanitizers are not available on the pool.
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../henjou-renderer_amd/host/frame.hpp"
#include "../../henjou-renderer_amd/host/scene.hpp"
#include "../../oracle/hjr_oracle.h"

namespace hjr {
void set_error(const std::string&) {}
bool write_png(const std::string& path, const uint8_t* rgba, uint32_t w, uint32_t h, bool flip_y, std::string& err);
bool read_png_rgba8(const std::string& path, std::vector<uint8_t>& rgba, int& w, int& h, std::string& err);
void float4_to_srgb8(const float* rgba, uint8_t* out, uint32_t n);
void tonemap_to_srgb8(const float* rgba, uint8_t* out, uint32_t n, int mode);
bool read_jpeg_rgba8(const std::vector<uint8_t>& file, std::vector<uint8_t>& rgba, int& w, int& h, std::string& err);
}

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "CHECK failed: %s (line %d)\n", #c, __LINE__); return 1; } } while (0)

int main(int argc, char** argv)
{
    CHECK(argc >= 4);
    const std::string assets = argv[1], config = argv[2], tmp = argv[3];
    std::string err;
    hjr_render_option opt;
    HJR_INIT(opt);
    CHECK(hjr::load_render_option(assets + "/" + config, opt, err));
    hjr::SceneData sc;
    CHECK(hjr::load_gltf(assets + "/" + opt.gltf_path, opt.gltf_name, sc, opt, err));
    const uint32_t ninst = (uint32_t)sc.instances.size();
    std::vector<float> m(ninst * 12), inv(ninst * 12);
    hjr::eval_transforms(sc, 1.0f / 24.0f, m.data(), inv.data());
    hjr_camera cam;
    hjr::eval_camera(sc, opt, 1.0f / 24.0f, cam);

    { // keyframe sampling rules of renderer/animation.h:42-67 on a hand-made track (keys 0.5, 1, 2 -> values 10, 20, 40)
        hjr::Track<hjr::float3_> tr;
        tr.push(0.5f, hjr::float3_{ 10, 0, 0 }); tr.push(1.0f, hjr::float3_{ 20, 0, 0 }); tr.push(2.0f, hjr::float3_{ 40, 0, 0 });
        CHECK(tr.at(-1.0f).x == 10.0f);  // negative time: first key
        CHECK(tr.at(0.25f).x == 40.0f);  // 0 <= time < first key: the reference's offset == -1 wraps and selects the LAST key (sic)
        CHECK(tr.at(0.5f).x == 10.0f && tr.at(0.75f).x == 15.0f && tr.at(1.0f).x == 20.0f && tr.at(1.5f).x == 30.0f);
        CHECK(tr.at(2.0f).x == 40.0f && tr.at(7.0f).x == 40.0f); // at / past the last key
        hjr::Track<hjr::float3_> one;
        one.push(3.0f, hjr::float3_{ 5, 6, 7 });
        CHECK(one.at(0.0f).y == 6.0f && one.at(9.0f).z == 7.0f); // a single key is constant
        hjr::NodeMotion nm; // identity TRS -> identity 3x4
        nm.translation.push(0, hjr::float3_{ 0, 0, 0 }); nm.rotation.push(0, hjr::float4_{ 0, 0, 0, 1 }); nm.scale.push(0, hjr::float3_{ 1, 1, 1 });
        float id[12];
        nm.matrix3x4(0.0f, id);
        for (int k = 0; k < 12; k++) CHECK(id[k] == ((k % 5 == 0) ? 1.0f : 0.0f));
    }

    hjr_scene_view v;
    HJR_INIT(v);
    v.n_vertices = (uint32_t)sc.vertices.size(); v.n_triangles = (uint32_t)sc.indices.size() / 3; v.n_instances = ninst;
    v.n_materials = (uint32_t)sc.materials.size(); v.n_lights = (uint32_t)sc.light_prim_ids.size();
    v.vertices = &sc.vertices[0].x; v.normals = &sc.normals[0].x; v.texcoords = &sc.texcoords[0].x;
    v.indices = sc.indices.data(); v.material_ids = sc.material_ids.data(); v.prim_offset = sc.prim_offset.data();
    v.materials = sc.materials.data(); v.light_prim_ids = sc.light_prim_ids.data();
    v.light_prim_emission = sc.light_prim_emission.empty() ? nullptr : &sc.light_prim_emission[0].x;
    v.n_textures = (uint32_t)sc.texture_views.size(); v.textures = sc.texture_views.data();
    hjr::SceneCopy copy;
    CHECK(copy.set(v, err));
    for (int allow_lds = 0; allow_lds < 2; allow_lds++) {
        hjr::FrameData fd;
        hjr::BuildOptions bo;
        bo.allow_lds = allow_lds != 0;
        CHECK(hjr::build_frame(copy, m.data(), inv.data(), ninst, bo, fd, err));
        CHECK(fd.n_tris == v.n_triangles && fd.n_nodes > 0 && fd.stack_need >= 2);
        CHECK(fd.width == (allow_lds ? 2u : 4u));
    }
    { // the insertion-based refinement (host/frame.cpp::Refine): same triangle set, a different tree, under the sanitizers
        hjr::FrameData plain, refined;
        hjr::BuildOptions bo;
        bo.refine = 0;
        CHECK(hjr::build_frame(copy, m.data(), inv.data(), ninst, bo, plain, err));
        bo.refine = 4;
        CHECK(hjr::build_frame(copy, m.data(), inv.data(), ninst, bo, refined, err));
        CHECK(refined.n_tris == plain.n_tris && refined.n_nodes == plain.n_nodes && refined.tri_geom == plain.tri_geom); // leaves (triangle order) untouched
        CHECK(refined.stack_need >= 2 && refined.depth < 32);
        CHECK(refined.nodes != plain.nodes); // (the bundled scenes do move subtrees)
    }

    // oracle render of a small frame through the same arrays
    std::vector<hjo_material> om(sc.materials.size());
    static_assert(sizeof(hjo_material) == sizeof(hjr_material), "material mirrors");
    memcpy(om.data(), sc.materials.data(), om.size() * sizeof(hjo_material));
    std::vector<hjo_texture> ot;
    for (auto& t : sc.texture_views) ot.push_back(hjo_texture{ t.rgba8, t.width, t.height, t.srgb, 0 });
    hjo_scene os;
    memset(&os, 0, sizeof(os));
    os.n_tris = v.n_triangles; os.n_instances = ninst; os.n_materials = v.n_materials; os.n_lights = v.n_lights;
    os.vertices = v.vertices; os.normals = v.normals; os.texcoords = v.texcoords; os.indices = v.indices;
    os.material_ids = v.material_ids; os.prim_offsets = v.prim_offset; os.transforms = m.data(); os.inv_transforms = inv.data();
    os.materials = om.data(); os.light_prim_ids = v.light_prim_ids; os.light_prim_emission = v.light_prim_emission;
    os.textures = ot.data(); os.n_textures = (uint32_t)ot.size();
    for (int mode = 0; mode < 2; mode++) {
        hjo_ctx* c = hjo_create(&os, mode);
        hjo_params p;
        memset(&p, 0, sizeof(p));
        p.width = 24; p.height = 16; p.spp = 3; p.frame = 1; p.seed = 1;
        memcpy(p.cam_pos, cam.pos, 12); memcpy(p.cam_dir, cam.dir, 12); memcpy(p.cam_up, cam.up, 12); memcpy(p.cam_right, cam.right, 12);
        p.cam_f = cam.f; p.sky[0] = p.sky[1] = p.sky[2] = 0.8f; p.ibl_intensity = 1.0f;
        std::vector<float> col(24 * 16 * 4), alb(24 * 16 * 4), nor(24 * 16 * 4);
        for (uint32_t integ = 0; integ < 3; integ++) {
            p.integrator = integ;
            hjo_stats st;
            CHECK(hjo_render(c, &p, col.data(), alb.data(), nor.data(), 2, &st) == 0);
            CHECK(st.samples == 24 * 16 * 3 && st.nan_samples == 0);
        }
        std::vector<uint8_t> px(24 * 16 * 4), back;
        hjr::tonemap_to_srgb8(col.data(), px.data(), 24 * 16, 1);
        hjr::float4_to_srgb8(col.data(), px.data(), 24 * 16);
        CHECK(hjr::write_png(tmp + "/san.png", px.data(), 24, 16, true, err));
        int w = 0, h = 0;
        CHECK(hjr::read_png_rgba8(tmp + "/san.png", back, w, h, err) && w == 24 && h == 16);
        hjo_destroy(c);
    }
    // JPEG decoder on a valid stream and on corrupted copies of it (tmp/fuzz.jpg is written by the test)
    if (FILE* f = fopen((tmp + "/fuzz.jpg").c_str(), "rb")) {
        std::vector<uint8_t> raw;
        int ch;
        while ((ch = fgetc(f)) != EOF) raw.push_back((uint8_t)ch);
        fclose(f);
        std::vector<uint8_t> px;
        int w = 0, h = 0;
        CHECK(hjr::read_jpeg_rgba8(raw, px, w, h, err) && w > 0 && h > 0 && px.size() == (size_t)w * h * 4);
        uint32_t lcg = 12345u;
        auto rnd = [&](uint32_t n) { lcg = lcg * 1664525u + 1013904223u; return (lcg >> 8) % n; };
        for (int trial = 0; trial < 400; trial++) {
            std::vector<uint8_t> b = raw;
            if (trial % 4 == 0) b.resize(2 + rnd((uint32_t)b.size() - 2));
            else for (uint32_t k = 0, n = 1 + rnd(6); k < n; k++) b[2 + rnd((uint32_t)b.size() - 2)] = (uint8_t)rnd(256);
            std::string e2;
            (void)hjr::read_jpeg_rgba8(b, px, w, h, e2);
        }
    }
    printf("sanitize_driver ok\n");
    return 0;
}
