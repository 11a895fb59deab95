// tools/kbench — kernel-variant bench over the C-ABI, no Python / torch start-up cost.
//   kbench <libhenjou_hip.so> <render_option.json> [--width W] [--height H] [--spp S] [--integrator 0|1|2] [--reps N]
//          [--aovs] [--rank r --world n] [--stats] [--fast] [--opt key=value ...]
// dlopens the given build of the library (so several builds can be compared inside one gpurun call), renders the start frame of
// the config `reps` times through hjr_render (host buffers; the kernel time is the library's own HIP-event time around the
// kernels) and prints kernel ms (min / mean), Msamples/s and an FNV-1a hash of the colour AOV (equal hashes = bit-identical
// frames).  Run it from the directory the config's relative paths refer to (henjou-renderer_amd/assets for the bundled ones).
// Build: g++ -O2 -std=c++17 tools/kbench.cpp -o tools/kbench -ldl
#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include "../include/henjou_hip.h"

#define SYM(name) auto p_##name = (decltype(&name))dlsym(h, #name); if (!p_##name) { fprintf(stderr, "kbench: missing symbol %s\n", #name); return 2; }

int main(int argc, char** argv)
{
    if (argc < 3) { fprintf(stderr, "usage: kbench <lib.so> <render_option.json> [options]\n"); return 2; }
    void* h = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!h) { fprintf(stderr, "kbench: %s\n", dlerror()); return 2; }
    SYM(hjr_last_error) SYM(hjr_load_render_option) SYM(hjr_scene_load_gltf) SYM(hjr_scene_get_view) SYM(hjr_scene_eval_transforms)
    SYM(hjr_scene_eval_camera) SYM(hjr_load_png_rgba8) SYM(hjr_free) SYM(hjr_create) SYM(hjr_upload_scene) SYM(hjr_set_transforms)
    SYM(hjr_set_lut) SYM(hjr_render) SYM(hjr_get_stats) SYM(hjr_destroy) SYM(hjr_set_option)
    hjr_render_option opt;
    HJR_INIT(opt);
#define CHK(call) do { int rc_ = (call); if (rc_ != HJR_OK) { fprintf(stderr, "kbench: %s -> %d: %s\n", #call, rc_, p_hjr_last_error()); return 1; } } while (0)
    CHK(p_hjr_load_render_option(argv[2], &opt));
    int reps = 3, rank = 0, world = 1;
    bool aovs = false, stats = false, fast = false;
    std::vector<std::pair<std::string, int>> options; // --opt key=value -> hjr_set_option (before the frame data is built)
    for (int i = 3; i < argc; i++) {
        std::string a = argv[i];
        auto val = [&]() { return (i + 1 < argc) ? atoi(argv[++i]) : 0; };
        if (a == "--width") opt.image_width = (uint32_t)val();
        else if (a == "--height") opt.image_height = (uint32_t)val();
        else if (a == "--spp") opt.max_spp = (uint32_t)val();
        else if (a == "--integrator") opt.integrator = val();
        else if (a == "--reps") reps = val();
        else if (a == "--rank") rank = val();
        else if (a == "--world") world = val();
        else if (a == "--aovs") aovs = true;
        else if (a == "--stats") stats = true;
        else if (a == "--fast") fast = true;
        else if (a == "--opt" && i + 1 < argc) { std::string kv = argv[++i]; const size_t eq = kv.find('='); if (eq == std::string::npos) { fprintf(stderr, "kbench: --opt key=value\n"); return 2; } options.push_back({ kv.substr(0, eq), atoi(kv.c_str() + eq + 1) }); }
        else { fprintf(stderr, "kbench: unknown option %s\n", a.c_str()); return 2; }
    }
    hjr_scene* scene = nullptr;
    CHK(p_hjr_scene_load_gltf(opt.gltf_path, opt.gltf_name, &opt, &scene));
    hjr_scene_view view;
    HJR_INIT(view);
    CHK(p_hjr_scene_get_view(scene, &view));
    hjr_ctx* ctx = nullptr;
    CHK(p_hjr_create(0, &ctx));
    for (auto& o : options) CHK(p_hjr_set_option(ctx, o.first.c_str(), o.second));
    CHK(p_hjr_upload_scene(ctx, &view));
    {
        uint8_t* lut = nullptr; int lw = 0, lh = 0;
        if (p_hjr_load_png_rgba8(opt.LUT_path, &lut, &lw, &lh) == HJR_OK) { CHK(p_hjr_set_lut(ctx, lut, lw, lh)); p_hjr_free(lut); }
    }
    const float time = opt.start_frame / float(opt.fps);
    std::vector<float> m((size_t)view.n_instances * 12), inv((size_t)view.n_instances * 12);
    CHK(p_hjr_scene_eval_transforms(scene, time, m.data(), inv.data()));
    CHK(p_hjr_set_transforms(ctx, m.data(), inv.data(), view.n_instances));
    hjr_params p;
    HJR_INIT(p);
    p.width = opt.image_width; p.height = opt.image_height; p.spp = opt.max_spp; p.frame = opt.start_frame; p.seed = opt.seed;
    p.integrator = (uint32_t)opt.integrator;
    CHK(p_hjr_scene_eval_camera(scene, &opt, time, &p.camera));
    for (int k = 0; k < 3; k++) p.sky[k] = opt.scene_sky_default[k];
    p.ibl_intensity = opt.IBL_intensity;
    p.rank = (uint32_t)rank; p.world_size = (uint32_t)world;
    p.flags = (stats ? HJR_FLAG_STATS : 0u) | (world > 1 ? HJR_FLAG_PACKED : 0u) | (fast ? HJR_FLAG_FAST_MATH : 0u); // a rank of a shard renders packed tiles, as bench.py and henjou_cli do
    const size_t npx = (size_t)p.width * p.height;
    std::vector<float> color(npx * 4), albedo(aovs ? npx * 4 : 0), normal(aovs ? npx * 4 : 0);
    double sum = 0, best = 1e30;
    hjr_stats st;
    HJR_INIT(st);
    for (int r = 0; r < reps + 1; r++) { // first launch is the warm-up
        CHK(p_hjr_render(ctx, &p, color.data(), aovs ? albedo.data() : nullptr, aovs ? normal.data() : nullptr));
        CHK(p_hjr_get_stats(ctx, &st));
        if (r == 0) continue;
        sum += st.last_kernel_ms;
        if (st.last_kernel_ms < best) best = st.last_kernel_ms;
    }
    auto fnv = [](const std::vector<float>& v) {
        uint64_t x = 1469598103934665603ull;
        const unsigned char* b = (const unsigned char*)v.data();
        for (size_t i = 0; i < v.size() * 4; i++) { x ^= b[i]; x *= 1099511628211ull; }
        return x;
    };
    const double samples = (double)npx * p.spp / world;
    printf("%s %ux%ux%u integ %d aovs %d rank %d/%d: kernel ms min %.3f mean %.3f  Msamples/s(min) %.1f  hash %016llx", argv[1], p.width, p.height,
           p.spp, opt.integrator, (int)aovs, rank, world, best, sum / reps, samples / (best * 1e3), (unsigned long long)fnv(color));
    if (aovs) printf(" %016llx %016llx", (unsigned long long)fnv(albedo), (unsigned long long)fnv(normal));
    printf("\n");
    if (stats) printf("  samples %llu closest %llu shadow %llu box_c %llu tri_c %llu box_s %llu tri_s %llu hits %llu lights %llu nan %llu\n",
                      (unsigned long long)st.samples, (unsigned long long)st.closest_rays, (unsigned long long)st.shadow_rays, (unsigned long long)st.box_tests_closest,
                      (unsigned long long)st.tri_tests_closest, (unsigned long long)st.box_tests_shadow, (unsigned long long)st.tri_tests_shadow,
                      (unsigned long long)st.shaded_hits, (unsigned long long)st.light_samples, (unsigned long long)st.nan_samples);
    if (stats) for (uint32_t i = 0; i < st.nan_located; i++) printf("  nan sample at x %u y %u s %u\n", st.nan_where[i][0], st.nan_where[i][1], st.nan_where[i][2]);
    p_hjr_destroy(ctx);
    return 0;
}
