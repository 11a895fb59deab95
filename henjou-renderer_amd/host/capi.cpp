// Host half of the C-ABI (include/henjou_hip.h): scene surface, output stage and the whole-file driver
// hjr_render_file == Renderer::initializeAndRender (renderer/renderer.h:1053-1317).
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/henjou_hip.h"
#include "../csrc/hjr_layout.h"
#include "abi.hpp"
#include "scene.hpp"

namespace hjr {
static thread_local std::string g_err;
void set_error(const std::string& s) { g_err = s; }
bool write_png(const std::string& path, const uint8_t* rgba, uint32_t w, uint32_t h, bool flip_y, std::string& err);
bool read_png_rgba8(const std::string& path, std::vector<uint8_t>& rgba, int& w, int& h, std::string& err);
bool read_image_rgba8(const std::string& path, std::vector<uint8_t>& rgba, int& w, int& h, std::string& err);
bool write_pfm(const std::string& path, const float* rgba, uint32_t w, uint32_t h, std::string& err);
bool read_hdr_rgba32f(const std::string& path, std::vector<float>& rgba, int& w, int& h, std::string& err);
void float4_to_srgb8(const float* rgba, uint8_t* out, uint32_t n);
void tonemap_to_srgb8(const float* rgba, uint8_t* out, uint32_t n, int mode);
} // namespace hjr
using hjr::set_error;

struct hjr_scene {
    hjr::SceneData data;
};

extern "C" const char* hjr_last_error(void) { return hjr::g_err.c_str(); }

extern "C" int hjr_load_render_option(const char* json_path, hjr_render_option* out)
{
    if (!json_path || !out) { set_error("hjr_load_render_option: null argument"); return HJR_ERR_ARG; }
    uint32_t out_size;
    if (!hjr::abi_size(out, out_size, "hjr_load_render_option")) return HJR_ERR_ARG;
    std::string err;
    hjr_render_option opt;
    if (!hjr::load_render_option(json_path, opt, err)) {
        set_error(err);
        return err.rfind("File ", 0) == 0 ? HJR_ERR_IO : HJR_ERR_PARSE;
    }
    opt.struct_size = (uint32_t)sizeof(opt);
    return hjr::abi_give(out, opt, "hjr_load_render_option") ? HJR_OK : HJR_ERR_ARG; // sized struct: at most out->struct_size bytes are written
}

extern "C" int hjr_scene_load_gltf(const char* dir, const char* file, hjr_render_option* opt, hjr_scene** out)
{
    if (!dir || !file || !opt || !out) { set_error("hjr_scene_load_gltf: null argument"); return HJR_ERR_ARG; }
    *out = nullptr;
    hjr_render_option o; // sized struct, in / out: the loader may enable the camera animation (gltfloader.h:1296-1310)
    if (!hjr::abi_take(opt, o, "hjr_scene_load_gltf")) return HJR_ERR_ARG;
    const std::string dir_s = dir, file_s = file; // (dir / file usually point into *opt)
    hjr_scene* s = new hjr_scene();
    std::string err;
    if (!hjr::load_gltf(dir_s, file_s, s->data, o, err)) {
        set_error(err);
        delete s;
        return err.find("cannot open") != std::string::npos ? HJR_ERR_IO : HJR_ERR_PARSE;
    }
    if (!hjr::abi_give(opt, o, "hjr_scene_load_gltf")) { delete s; return HJR_ERR_ARG; }
    *out = s;
    return HJR_OK;
}

extern "C" void hjr_scene_free(hjr_scene* s) { delete s; }

extern "C" int hjr_scene_get_view(const hjr_scene* s, hjr_scene_view* v)
{
    if (!s || !v) { set_error("hjr_scene_get_view: null argument"); return HJR_ERR_ARG; }
    hjr_scene_view* const user = v;
    hjr_scene_view full; // sized struct: filled here, at most user->struct_size bytes handed out
    v = &full;
    const hjr::SceneData& d = s->data;
    memset(v, 0, sizeof(*v));
    v->struct_size = (uint32_t)sizeof(*v);
    v->n_vertices = (uint32_t)d.vertices.size();
    v->n_triangles = (uint32_t)(d.indices.size() / 3);
    v->n_instances = (uint32_t)d.instances.size();
    v->n_materials = (uint32_t)d.materials.size();
    v->n_lights = (uint32_t)d.light_prim_ids.size();
    v->n_animations = (uint32_t)d.animations.size();
    v->vertices = d.vertices.empty() ? nullptr : &d.vertices[0].x;
    v->normals = d.normals.empty() ? nullptr : &d.normals[0].x;
    v->texcoords = d.texcoords.empty() ? nullptr : &d.texcoords[0].x;
    v->indices = d.indices.data();
    v->material_ids = d.material_ids.data();
    v->prim_offset = d.prim_offset.data();
    v->geometry_index_offset = d.geo_index_offset.data();
    v->geometry_index_count = d.geo_index_count.data();
    v->instance_animation_id = d.inst_animation_id.data();
    v->materials = d.materials.data();
    v->light_prim_ids = d.light_prim_ids.data();
    v->light_prim_emission = d.light_prim_emission.empty() ? nullptr : &d.light_prim_emission[0].x;
    v->n_textures = (uint32_t)d.texture_views.size();
    v->textures = d.texture_views.data();
    return hjr::abi_give(user, full, "hjr_scene_get_view") ? HJR_OK : HJR_ERR_ARG;
}

extern "C" int hjr_scene_eval_transforms(const hjr_scene* s, float time, float* m12, float* inv12)
{
    if (!s || ((!m12 || !inv12) && !s->data.instances.empty())) { set_error("hjr_scene_eval_transforms: null argument"); return HJR_ERR_ARG; }
    hjr::eval_transforms(s->data, time, m12, inv12);
    return HJR_OK;
}

extern "C" int hjr_scene_eval_camera(const hjr_scene* s, const hjr_render_option* opt, float time, hjr_camera* out)
{
    if (!s || !opt || !out) { set_error("hjr_scene_eval_camera: null argument"); return HJR_ERR_ARG; }
    hjr_render_option o; // sized struct
    if (!hjr::abi_take(opt, o, "hjr_scene_eval_camera")) return HJR_ERR_ARG;
    hjr::eval_camera(s->data, o, time, *out);
    return HJR_OK;
}

extern "C" int hjr_load_png_rgba8(const char* path, uint8_t** rgba, int* w, int* h)
{
    if (!path || !rgba || !w || !h) { set_error("hjr_load_png_rgba8: null argument"); return HJR_ERR_ARG; }
    std::vector<uint8_t> px;
    std::string err;
    if (!hjr::read_png_rgba8(path, px, *w, *h, err)) {
        set_error(err);
        return err.rfind("cannot open", 0) == 0 ? HJR_ERR_IO : HJR_ERR_PARSE;
    }
    *rgba = (uint8_t*)malloc(px.size());
    if (!*rgba) { set_error("out of memory"); return HJR_ERR_ARG; }
    memcpy(*rgba, px.data(), px.size());
    return HJR_OK;
}

extern "C" int hjr_load_image_rgba8(const char* path, uint8_t** rgba, int* w, int* h)
{
    if (!path || !rgba || !w || !h) { set_error("hjr_load_image_rgba8: null argument"); return HJR_ERR_ARG; }
    std::vector<uint8_t> px;
    std::string err;
    if (!hjr::read_image_rgba8(path, px, *w, *h, err)) {
        set_error(err);
        return err.rfind("cannot open", 0) == 0 ? HJR_ERR_IO : HJR_ERR_PARSE;
    }
    *rgba = (uint8_t*)malloc(px.size());
    if (!*rgba) { set_error("out of memory"); return HJR_ERR_ARG; }
    memcpy(*rgba, px.data(), px.size());
    return HJR_OK;
}

extern "C" int hjr_load_hdr_rgba32f(const char* path, float** rgba, int* w, int* h)
{
    if (!path || !rgba || !w || !h) { set_error("hjr_load_hdr_rgba32f: null argument"); return HJR_ERR_ARG; }
    std::vector<float> px;
    std::string err;
    if (!hjr::read_hdr_rgba32f(path, px, *w, *h, err)) {
        set_error(err);
        return err.rfind("cannot open", 0) == 0 ? HJR_ERR_IO : HJR_ERR_PARSE;
    }
    *rgba = (float*)malloc(px.size() * sizeof(float));
    if (!*rgba) { set_error("out of memory"); return HJR_ERR_ARG; }
    memcpy(*rgba, px.data(), px.size() * sizeof(float));
    return HJR_OK;
}

extern "C" void hjr_free(void* p) { free(p); }

// ---- pixel-tile shard: 8x8 tiles, tile t -> rank t % world; packed form = the rank's tiles back to back
extern "C" uint32_t hjr_owned_tiles(uint32_t w, uint32_t h, uint32_t rank, uint32_t world)
{
    if (world == 0 || rank >= world) return 0;
    const uint64_t n_tiles = (uint64_t)((w + 7u) / 8u) * ((h + 7u) / 8u);
    return n_tiles > rank ? (uint32_t)((n_tiles - rank + world - 1) / world) : 0u;
}
template <bool PACK> static int tiles_copy(const float* src, float* dst, uint32_t w, uint32_t h, uint32_t rank, uint32_t world)
{
    if (!src || !dst || w == 0 || h == 0 || world == 0 || rank >= world) { set_error("hjr_pack_tiles / hjr_unpack_tiles: bad argument"); return HJR_ERR_ARG; }
    const uint32_t tiles_x = (w + 7u) / 8u, n = hjr_owned_tiles(w, h, rank, world);
    for (uint32_t i = 0; i < n; i++) {
        uint32_t tx, ty;
        hjr_tile_xy(i * world + rank, tiles_x, &tx, &ty);
        const uint32_t x0 = tx * 8u, y0 = ty * 8u;
        for (uint32_t k = 0; k < 64u; k++) {
            const uint32_t x = x0 + (k & 7u), y = y0 + (k >> 3);
            float* p = PACK ? dst + ((size_t)i * 64 + k) * 4 : nullptr;
            if (x < w && y < h) {
                const size_t f = ((size_t)y * w + x) * 4, q = ((size_t)i * 64 + k) * 4;
                if (PACK) memcpy(dst + q, src + f, 16); else memcpy(dst + f, src + q, 16);
            } else if (PACK) p[0] = p[1] = p[2] = p[3] = 0.0f;
        }
    }
    return HJR_OK;
}
extern "C" int hjr_pack_tiles(const float* frame, uint32_t w, uint32_t h, uint32_t rank, uint32_t world, float* packed) { return tiles_copy<true>(frame, packed, w, h, rank, world); }
extern "C" int hjr_unpack_tiles(const float* packed, uint32_t w, uint32_t h, uint32_t rank, uint32_t world, float* frame) { return tiles_copy<false>(packed, frame, w, h, rank, world); }

extern "C" int hjr_float4_to_srgb8(const float* rgba, uint8_t* out, uint32_t n)
{
    if ((!rgba || !out) && n) { set_error("hjr_float4_to_srgb8: null argument"); return HJR_ERR_ARG; }
    hjr::float4_to_srgb8(rgba, out, n);
    return HJR_OK;
}

extern "C" int hjr_tonemap_to_srgb8(const float* rgba, uint8_t* out, uint32_t n, int tonemap)
{
    if ((!rgba || !out) && n) { set_error("hjr_tonemap_to_srgb8: null argument"); return HJR_ERR_ARG; }
    if (tonemap < HJR_TONEMAP_NONE || tonemap > HJR_TONEMAP_ACES) { set_error("hjr_tonemap_to_srgb8: unknown tonemap"); return HJR_ERR_ARG; }
    hjr::tonemap_to_srgb8(rgba, out, n, tonemap);
    return HJR_OK;
}

extern "C" int hjr_write_png(const char* path, const uint8_t* rgba8, uint32_t w, uint32_t h, int flip_y)
{
    if (!path || !rgba8) { set_error("hjr_write_png: null argument"); return HJR_ERR_ARG; }
    std::string err;
    if (!hjr::write_png(path, rgba8, w, h, flip_y != 0, err)) { set_error(err); return HJR_ERR_IO; }
    return HJR_OK;
}

extern "C" int hjr_write_pfm(const char* path, const float* rgba, uint32_t w, uint32_t h)
{
    if (!path || !rgba) { set_error("hjr_write_pfm: null argument"); return HJR_ERR_ARG; }
    std::string err;
    if (!hjr::write_pfm(path, rgba, w, h, err)) { set_error(err); return HJR_ERR_IO; }
    return HJR_OK;
}

// Renderer::initializeAndRender — renderer/renderer.h:1053-1317.  Render_mode "Default" is the pass-through of the reference
// (its denoiser runs with blendFactor 1); "Denoise" / "DenoiseUpScale2X" keep the reference's data flow with the HIP a-trous
// filter in place of the closed OptiX network (csrc/hjr_denoise.hip.h).
extern "C" int hjr_render_file(const char* render_option_json, int device)
{
    if (!render_option_json) { set_error("hjr_render_file: null path"); return HJR_ERR_ARG; }
    hjr_render_option opt;
    HJR_INIT(opt);
    int rc = hjr_load_render_option(render_option_json, &opt);
    if (rc != HJR_OK) return rc;
    if (opt.render_mode != HJR_MODE_DEFAULT && opt.render_mode != HJR_MODE_DENOISE && opt.render_mode != HJR_MODE_DENOISE_UPSCALE2X) {
        set_error("hjr_render_file: Render_mode must be Default, Denoise or DenoiseUpScale2X (Debug is declared but unused by the reference)");
        return HJR_ERR_ARG;
    }
    hjr_scene* scene = nullptr;
    rc = hjr_scene_load_gltf(opt.gltf_path, opt.gltf_name, &opt, &scene);
    if (rc != HJR_OK) return rc;
    hjr_ctx* ctx = nullptr;
    rc = hjr_create(device, &ctx);
    if (rc != HJR_OK) { hjr_scene_free(scene); return rc; }
    hjr_scene_view view;
    HJR_INIT(view);
    hjr_scene_get_view(scene, &view);
    if (opt.force_rebuild) (void)hjr_set_option(ctx, "force_rebuild", 1);
    rc = hjr_upload_scene(ctx, &view);
    if (rc == HJR_OK) { // setLUT (renderer.h:854-898); a missing LUT file only matters if a material uses it
        uint8_t* lut = nullptr;
        int lw = 0, lh = 0;
        if (hjr_load_png_rgba8(opt.LUT_path, &lut, &lw, &lh) == HJR_OK) {
            rc = hjr_set_lut(ctx, lut, lw, lh);
            hjr_free(lut);
        } else {
            bool needs = false;
            for (uint32_t i = 0; i < view.n_materials; i++) needs = needs || view.materials[i].is_thinfilm;
            if (needs) rc = HJR_ERR_IO; // hjr_last_error() already holds the PNG error
        }
    }
    if (rc == HJR_OK && opt.use_IBL) { // setSky (renderer.h:802-851): a missing / undecodable HDR falls back to the 1x1 scene_sky_default texel (texture.h:89-98)
        float* sky = nullptr;
        int sw = 0, sh = 0;
        if (hjr_load_hdr_rgba32f(opt.IBL_path, &sky, &sw, &sh) == HJR_OK) {
            rc = hjr_set_sky(ctx, sky, sw, sh);
            hjr_free(sky);
        } else fprintf(stderr, "[henjou] %s NOT FOUND: using scene_sky_default\n", opt.IBL_path);
    }
    std::vector<float> m((size_t)view.n_instances * 12), inv((size_t)view.n_instances * 12);
    // Image Scale Setting (renderer.h:1089-1099): DenoiseUpScale2X renders at half the output size
    const uint32_t in_w = opt.render_mode == HJR_MODE_DENOISE_UPSCALE2X ? opt.image_width / 2u : opt.image_width;
    const uint32_t in_h = opt.render_mode == HJR_MODE_DENOISE_UPSCALE2X ? opt.image_height / 2u : opt.image_height;
    if (in_w == 0 || in_h == 0) { set_error("hjr_render_file: image too small for DenoiseUpScale2X"); hjr_destroy(ctx); hjr_scene_free(scene); return HJR_ERR_ARG; }
    const size_t npx = (size_t)opt.image_width * opt.image_height;
    // Output stage off the critical path: float4 -> sRGB8 -> PNG -> file runs on a writer thread while the main thread
    // already builds and renders the next frame (two frame buffers in rotation).  The reference's loop is serial
    // (renderer.h:1281-1302); the files are the same.  Only aov_color is produced: Default mode never reads the albedo /
    // normal AOVs (they feed the OptiX denoiser, denoiser.h:94-97).  "Henjou_HIP": {"serial_io": true} disables the overlap.
    struct Slot { std::vector<float> color; std::string name; bool full = false; };
    Slot slots[2];
    for (Slot& sl : slots) sl.color.resize(npx * 4);
    std::mutex mu;
    std::condition_variable cv;
    bool quit = false;
    int write_rc = HJR_OK;
    std::string write_err;
    const bool serial_io = opt.serial_io != 0;
    auto write_slot = [&](Slot& sl) -> int {
        std::vector<uint8_t> rgba8(npx * 4);
        hjr::float4_to_srgb8(sl.color.data(), rgba8.data(), (uint32_t)npx);
        std::string err;
        if (!hjr::write_png(sl.name, rgba8.data(), opt.image_width, opt.image_height, true, err)) {
            std::lock_guard<std::mutex> lk(mu);
            write_err = err;
            return HJR_ERR_IO;
        }
        return HJR_OK;
    };
    std::thread writer;
    if (!serial_io)
        writer = std::thread([&]() {
            int next = 0;
            for (;;) {
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return slots[next].full || quit; });
                    if (!slots[next].full) return; // quit with nothing pending
                }
                const int r = write_slot(slots[next]);
                {
                    std::lock_guard<std::mutex> lk(mu);
                    slots[next].full = false;
                    if (r != HJR_OK && write_rc == HJR_OK) write_rc = r;
                }
                cv.notify_all();
                next ^= 1;
            }
        });
    int cur = 0;
    const auto t_all0 = std::chrono::steady_clock::now();
    uint32_t n_frames = 0;
    // scene update of frame f + 1 (host: TRS evaluation, flatten, BVH build) runs on a helper thread while frame f renders;
    // only its upload (hjr_commit_transforms) waits for the render.  The reference's loop is serial (renderer.h:1128-1137).
    std::thread prep;
    int prep_rc = HJR_OK;
    std::string prep_err;
    auto prepare = [&](uint32_t frame) {
        float time = frame / float(opt.fps); // renderer.h:1128
        hjr_scene_eval_transforms(scene, time, m.data(), inv.data());
        prep_rc = hjr_prepare_transforms(ctx, m.data(), inv.data(), view.n_instances);
        if (prep_rc != HJR_OK) prep_err = hjr_last_error(); // thread-local on the helper thread
    };
    if (opt.start_frame < opt.end_frame) prepare(opt.start_frame);
    for (uint32_t frame = opt.start_frame; rc == HJR_OK && frame < opt.end_frame; frame++) {
        float time = frame / float(opt.fps); // renderer.h:1128
        if (prep.joinable()) prep.join();
        if (prep_rc != HJR_OK) { rc = prep_rc; set_error(prep_err); break; }
        rc = hjr_commit_transforms(ctx);
        if (rc != HJR_OK) break;
        if (frame + 1 < opt.end_frame && !serial_io) prep = std::thread(prepare, frame + 1);
        hjr_params p;
        HJR_INIT(p);
        p.width = in_w; p.height = in_h;
        p.spp = opt.max_spp; p.frame = frame; p.seed = opt.seed; p.integrator = (uint32_t)opt.integrator;
        hjr_scene_eval_camera(scene, &opt, time, &p.camera);
        for (int k = 0; k < 3; k++) p.sky[k] = opt.scene_sky_default[k];
        p.ibl_intensity = opt.IBL_intensity;
        p.rank = 0; p.world_size = 1;
        if (opt.fast_math) p.flags |= HJR_FLAG_FAST_MATH; // "Henjou_HIP": {"fast_math": true}
        Slot& sl = slots[cur];
        if (!serial_io) { // the slot may still be with the writer (two frames behind)
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return !sl.full; });
            if (write_rc != HJR_OK) { rc = write_rc; break; }
        }
        if (opt.render_mode == HJR_MODE_DEFAULT) rc = hjr_render(ctx, &p, sl.color.data(), nullptr, nullptr);
        else rc = hjr_render_denoised(ctx, &p, opt.render_mode, sl.color.data(), opt.image_width, opt.image_height); // renderer.h:1258-1281
        if (rc != HJR_OK) break;
        hjr_stats st;
        HJR_INIT(st);
        if (hjr_get_stats(ctx, &st) == HJR_OK)
            fprintf(stderr, "[henjou] frame %u: %ux%u, %u spp, kernel %.3f ms (%.2f Msamples/s)\n", frame, p.width, p.height, p.spp,
                    st.last_kernel_ms, st.last_kernel_ms > 0 ? (double)p.width * p.height * p.spp / (st.last_kernel_ms * 1e3) : 0.0);
        std::string str_frame = std::to_string(frame); // renderer.h:1291-1302
        if (str_frame.size() < 2) str_frame = "00" + str_frame;
        else if (str_frame.size() < 3) str_frame = "0" + str_frame;
        sl.name = std::string(opt.image_name) + "_" + str_frame + ".png";
        n_frames++;
        if (serial_io) { rc = write_slot(sl); if (rc != HJR_OK) set_error(write_err); if (rc == HJR_OK && frame + 1 < opt.end_frame) prepare(frame + 1); }
        else {
            { std::lock_guard<std::mutex> lk(mu); sl.full = true; }
            cv.notify_all();
            cur ^= 1;
        }
    }
    if (prep.joinable()) prep.join();
    if (!serial_io) {
        { // drain: the writer takes the slots in order, then quits
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return !slots[0].full && !slots[1].full; });
            quit = true;
        }
        cv.notify_all();
        writer.join();
        if (rc == HJR_OK && write_rc != HJR_OK) { rc = write_rc; set_error(write_err); }
    }
    if (n_frames) {
        const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_all0).count();
        fprintf(stderr, "[henjou] %u frame(s) in %.3f s wall (%.1f ms per frame incl. scene update, download and PNG output)\n", n_frames, wall, 1e3 * wall / n_frames);
    }
    hjr_destroy(ctx);
    hjr_scene_free(scene);
    return rc;
}
