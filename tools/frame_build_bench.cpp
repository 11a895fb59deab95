// Host-side timing of hjr::build_frame (flatten + BVH build + emit) for a render_option.json scene.
//   g++ -O3 -std=c++17 -I. tools/frame_build_bench.cpp henjou-renderer_amd/host/{loaders,frame,image_io,jpeg}.cpp -lz -pthread -o /tmp/fbb
//   /tmp/fbb <dir containing render_option + Model/> <render_option.json> [repeats [threads [verbose [refine passes]]]]
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../henjou-renderer_amd/host/frame.hpp"
#include "../henjou-renderer_amd/host/scene.hpp"

namespace hjr { void set_error(const std::string&) {} }

int main(int argc, char** argv)
{
    if (argc < 3) return 2;
    const std::string dir = argv[1], config = argv[2];
    const int reps = argc > 3 ? atoi(argv[3]) : 3;
    if (argc > 4) hjr::set_host_threads(atoi(argv[4])); // worker threads (default min(hardware threads, 16))
    hjr::BuildOptions bo;
    bo.timing = argc > 5 && atoi(argv[5]) != 0;          // stage times on stderr
    if (argc > 6) bo.refine = atoi(argv[6]);             // option "bvh_refine" (passes; -1 default)
    if (argc > 7) bo.bvh_width = atoi(argv[7]);          // option "bvh_width"
    std::string err;
    hjr_render_option opt;
    HJR_INIT(opt);
    if (!hjr::load_render_option(dir + "/" + config, opt, err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
    hjr::SceneData sc;
    if (!hjr::load_gltf((opt.gltf_path[0] == 0x2f ? std::string() : dir + "/") + opt.gltf_path, opt.gltf_name, sc, opt, err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
    const uint32_t ninst = (uint32_t)sc.instances.size();
    std::vector<float> m(ninst * 12), inv(ninst * 12);
    hjr::eval_transforms(sc, 1.0f / 24.0f, m.data(), inv.data());
    hjr_scene_view v;
    HJR_INIT(v);
    v.n_vertices = (uint32_t)sc.vertices.size(); v.n_triangles = (uint32_t)sc.indices.size() / 3; v.n_instances = ninst;
    v.n_materials = (uint32_t)sc.materials.size(); v.n_lights = (uint32_t)sc.light_prim_ids.size();
    v.vertices = &sc.vertices[0].x; v.normals = &sc.normals[0].x; v.texcoords = &sc.texcoords[0].x;
    v.indices = sc.indices.data(); v.material_ids = sc.material_ids.data(); v.prim_offset = sc.prim_offset.data();
    v.materials = sc.materials.data(); v.light_prim_ids = sc.light_prim_ids.data();
    v.light_prim_emission = sc.light_prim_emission.empty() ? nullptr : &sc.light_prim_emission[0].x;
    hjr::SceneCopy copy;
    if (!copy.set(v, err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
    for (int r = 0; r < reps; r++) {
        hjr::FrameData fd;
        auto t0 = std::chrono::steady_clock::now();
        if (!hjr::build_frame(copy, m.data(), inv.data(), ninst, bo, fd, err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
        double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        unsigned long long h = 1469598103934665603ull; // FNV-1a over the emitted arrays: the build must not depend on threads
        auto mix = [&](const void* p, size_t n) { const unsigned char* b = (const unsigned char*)p; for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; } };
        mix(fd.nodes.data(), fd.nodes.size() * 4); mix(fd.tri_geom.data(), fd.tri_geom.size() * 4); mix(fd.tri_shade.data(), fd.tri_shade.size() * 4);
        printf("build_frame: %.1f ms  (%u tris, %u nodes, width %u, depth %u, stack %u, lds_mode %d)  hash %016llx\n", ms, fd.n_tris, fd.n_nodes, fd.width, fd.depth, fd.stack_need, fd.lds_mode, h);
    }
    return 0;
}
