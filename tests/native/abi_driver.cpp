// Sized structs of the C-ABI (include/henjou_hip.h, "Sized structs"): every entry point that takes hjr_render_option /
// hjr_scene_view / hjr_params / hjr_stats is called with a struct SHORTER than the library's (what a caller built against an
// older header owns) and with one LONGER (a newer caller), each on the heap with exactly struct_size bytes, so AddressSanitizer
// reports any byte the library reads or writes outside.  Built with -fsanitize=address,undefined by tests/test_abi_sizes.py from
// the product's own host sources; the device entry points that host/capi.cpp refers to are stubbed (no GPU code runs here), and
// the two structs that only cross the boundary on the device side (hjr_params in, hjr_stats out) go through the same
// host/abi.hpp functions the device code calls.
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../../henjou-renderer_amd/host/abi.hpp"
#include "../../include/henjou_hip.h"

extern "C" { // device half of the library: not under test here
int hjr_create(int, hjr_ctx**) { return HJR_ERR_DEVICE; }
void hjr_destroy(hjr_ctx*) {}
int hjr_upload_scene(hjr_ctx*, const hjr_scene_view*) { return HJR_ERR_DEVICE; }
int hjr_set_lut(hjr_ctx*, const uint8_t*, int, int) { return HJR_ERR_DEVICE; }
int hjr_set_sky(hjr_ctx*, const float*, int, int) { return HJR_ERR_DEVICE; }
int hjr_prepare_transforms(hjr_ctx*, const float*, const float*, uint32_t) { return HJR_ERR_DEVICE; }
int hjr_commit_transforms(hjr_ctx*) { return HJR_ERR_DEVICE; }
int hjr_render(hjr_ctx*, const hjr_params*, float*, float*, float*) { return HJR_ERR_DEVICE; }
int hjr_render_denoised(hjr_ctx*, const hjr_params*, int, float*, uint32_t, uint32_t) { return HJR_ERR_DEVICE; }
int hjr_get_stats(hjr_ctx*, hjr_stats*) { return HJR_ERR_DEVICE; }
int hjr_set_option(hjr_ctx*, const char*, int) { return HJR_ERR_DEVICE; }
}

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "CHECK failed: %s (line %d): %s\n", #c, __LINE__, hjr_last_error()); return 1; } } while (0)

// a struct of exactly `size` bytes on the heap, zero-filled, struct_size set
template <class T> static T* sized(size_t size, unsigned char fill = 0)
{
    T* p = (T*)malloc(size);
    memset((void*)p, fill, size);
    const uint32_t n = (uint32_t)size;
    memcpy((void*)p, &n, 4);
    return p;
}

int main(int argc, char** argv)
{
    CHECK(argc >= 3);
    const std::string assets = argv[1], config = assets + "/" + argv[2];
    hjr_render_option full;
    HJR_INIT(full);
    CHECK(hjr_load_render_option(config.c_str(), &full) == HJR_OK);
    CHECK(full.struct_size == sizeof(full) && full.image_width > 0 && full.devices == 1 && full.tile == 8);

    // ---- hjr_render_option, output: the round-1 layout ended before `seed` (the Henjou_HIP section came later)
    const size_t r1 = offsetof(hjr_render_option, seed);
    hjr_render_option* o_short = sized<hjr_render_option>(r1);
    CHECK(hjr_load_render_option(config.c_str(), o_short) == HJR_OK);
    CHECK(o_short->struct_size == r1 && memcmp(&o_short->image_width, &full.image_width, r1 - 4) == 0);
    // ... a struct that ends in the middle of a field
    hjr_render_option* o_odd = sized<hjr_render_option>(offsetof(hjr_render_option, gltf_path) + 5);
    CHECK(hjr_load_render_option(config.c_str(), o_odd) == HJR_OK && o_odd->max_spp == full.max_spp);
    free(o_odd);
    // ... a newer caller: the bytes this library does not know stay as they were
    hjr_render_option* o_long = sized<hjr_render_option>(sizeof(hjr_render_option) + 64, 0xAB);
    CHECK(hjr_load_render_option(config.c_str(), o_long) == HJR_OK);
    CHECK(o_long->struct_size == sizeof(hjr_render_option) + 64 && o_long->tile == 8);
    for (size_t i = 0; i < 64; i++) CHECK(((unsigned char*)o_long)[sizeof(hjr_render_option) + i] == 0xAB);
    free(o_long);
    // ... forgotten initialisation (zero-filled struct) is an error, not a guess
    hjr_render_option zero;
    memset(&zero, 0, sizeof(zero));
    CHECK(hjr_load_render_option(config.c_str(), &zero) == HJR_ERR_ARG);

    // ---- hjr_render_option, in / out (glTF loader) and in (camera): the short struct again; missing fields read as defaults
    hjr_scene* scene = nullptr;
    const std::string dir = assets + "/" + full.gltf_path;
    CHECK(hjr_scene_load_gltf(dir.c_str(), full.gltf_name, o_short, &scene) == HJR_OK && scene);
    {
        hjr_scene* again = nullptr; // the loader updates the camera fields of the option block: the full struct gets the same treatment
        CHECK(hjr_scene_load_gltf(dir.c_str(), full.gltf_name, &full, &again) == HJR_OK && again);
        hjr_scene_free(again);
        CHECK(memcmp(&o_short->image_width, &full.image_width, r1 - 4) == 0);
    }
    hjr_camera cam_s, cam_f;
    CHECK(hjr_scene_eval_camera(scene, o_short, 0.0f, &cam_s) == HJR_OK);
    CHECK(hjr_scene_eval_camera(scene, &full, 0.0f, &cam_f) == HJR_OK);
    CHECK(memcmp(&cam_s, &cam_f, sizeof(cam_s)) == 0);
    CHECK(hjr_scene_eval_camera(scene, &zero, 0.0f, &cam_s) == HJR_ERR_ARG);
    free(o_short);

    // ---- hjr_scene_view, output: a caller from before `textures` was appended
    hjr_scene_view vf;
    HJR_INIT(vf);
    CHECK(hjr_scene_get_view(scene, &vf) == HJR_OK && vf.n_triangles > 0 && vf.struct_size == sizeof(vf));
    const size_t v1 = offsetof(hjr_scene_view, textures);
    hjr_scene_view* v_short = sized<hjr_scene_view>(v1);
    CHECK(hjr_scene_get_view(scene, v_short) == HJR_OK);
    CHECK(v_short->struct_size == v1 && v_short->n_triangles == vf.n_triangles && v_short->light_prim_emission == vf.light_prim_emission);
    free(v_short);
    hjr_scene_view vz;
    memset(&vz, 0, sizeof(vz));
    CHECK(hjr_scene_get_view(scene, &vz) == HJR_ERR_ARG);
    hjr_scene_free(scene);

    // ---- hjr_params (input of the render entry points) and hjr_stats (output of hjr_get_stats) through the functions those call
    {
        const size_t p1 = offsetof(hjr_params, rank); // a caller from before the tile shard existed
        hjr_params* p_short = sized<hjr_params>(p1);
        p_short->width = 64; p_short->height = 32; p_short->spp = 4;
        hjr_params lp;
        CHECK(hjr::abi_take(p_short, lp, "test"));
        CHECK(lp.struct_size == sizeof(lp) && lp.width == 64 && lp.height == 32 && lp.spp == 4 && lp.rank == 0 && lp.world_size == 0 && lp.flags == 0);
        free(p_short);
        hjr_params pz;
        memset(&pz, 0, sizeof(pz));
        CHECK(!hjr::abi_take(&pz, lp, "test"));

        hjr_stats lib_stats;
        memset(&lib_stats, 0, sizeof(lib_stats));
        lib_stats.struct_size = sizeof(lib_stats);
        lib_stats.samples = 123; lib_stats.last_kernel_ms = 4.5f; lib_stats.lds_mode = 2; lib_stats.stack_overflow_pushes = 77;
        const size_t s1 = offsetof(hjr_stats, lds_mode); // the round-1 hjr_stats ended here: the struct tools/kbench had on its stack in round 2
        hjr_stats* s_short = sized<hjr_stats>(s1);
        CHECK(hjr::abi_give(s_short, lib_stats, "test"));
        CHECK(s_short->struct_size == s1 && s_short->samples == 123 && s_short->last_kernel_ms == 4.5f);
        free(s_short);
        hjr_stats* s_long = sized<hjr_stats>(sizeof(hjr_stats) + 32, 0xCD);
        CHECK(hjr::abi_give(s_long, lib_stats, "test") && s_long->stack_overflow_pushes == 77);
        for (size_t i = 0; i < 32; i++) CHECK(((unsigned char*)s_long)[sizeof(hjr_stats) + i] == 0xCD);
        free(s_long);
    }
    printf("abi_driver ok\n");
    return 0;
}
