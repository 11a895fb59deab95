// henjou_cli <render_option.json> [device] [--devices N]
// Stands in for the reference's missing main() (HenjouRenderer/henjouRenderer.cpp: Renderer r; r.initializeAndRender(path)).
//
// One GPU: hjr_render_file.  N GPUs of one node ("Henjou_HIP": {"devices": N} in the JSON, or --devices N): this process forks N
// rank processes BEFORE anything touches a GPU (fork + exec of itself with --rank), one per GPU; each loads the scene, renders its
// 8x8 pixel tiles of every frame (tile t -> rank t % N, HJR_FLAG_PACKED: only owned tiles are produced, [owned tile][64] float4)
// and the ranks meet in ONE RCCL collective per frame: ncclGather of the packed tiles onto rank 0 over xGMI (point-to-point sends,
// all peers in parallel, 1 / N of the frame per rank; SURVEY.md §8e option b).  Rank 0 scatters the N blocks into the frame
// (hjr_unpack_tiles_device), downloads it and writes <image_name>_<fff>.png exactly like the single-GPU path.  The assembled
// frame is bit-identical to the 1-GPU frame: per-pixel sample order does not depend on which GPU owns the pixel.
// The RCCL id travels over pipes the ranks inherit from the launcher (rank 0 -> launcher -> every other rank): no file, no name
// another user of the machine could guess.  A rank does everything that can fail (config, scene, device context, uploads) BEFORE it
// joins the communicator; the launcher reaps its children as they end, and the first one that fails or is signalled makes it
// terminate the others (SIGTERM, then SIGKILL), which may already be blocked inside ncclCommInitRank / ncclGather: a bad
// --devices value or a bad asset ends the job with exit code 1 instead of hanging it with the GPUs held.
// The reference has no counterpart (one process, one stream: renderer/renderer.h:1077-1078).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <poll.h>
#include <signal.h>
#include <sys/wait.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/henjou_hip.h"

#define HIPX(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { fprintf(stderr, "henjou_cli[%d]: %s: %s\n", rank, #call, hipGetErrorString(e_)); return 1; } } while (0)
#define NCCLX(call) do { ncclResult_t e_ = (call); if (e_ != ncclSuccess) { fprintf(stderr, "henjou_cli[%d]: %s: %s\n", rank, #call, ncclGetErrorString(e_)); return 1; } } while (0)
#define HJRX(call) do { int e_ = (call); if (e_ != HJR_OK) { fprintf(stderr, "henjou_cli[%d]: %s -> %d: %s\n", rank, #call, e_, hjr_last_error()); return 1; } } while (0)

// reads / writes exactly n bytes (false on EOF or error)
static bool io_all(int fd, void* buf, size_t n, bool write_it)
{
    char* b = (char*)buf;
    while (n) {
        const ssize_t k = write_it ? write(fd, b, n) : read(fd, b, n);
        if (k <= 0) return false;
        b += k; n -= (size_t)k;
    }
    return true;
}

// one rank of a multi-GPU render (its own process; GPU `rank` of the node)
static int run_rank(const char* json, int rank, int world, int id_fd)
{
    hjr_render_option opt;
    HJR_INIT(opt);
    HJRX(hjr_load_render_option(json, &opt));
    if (opt.render_mode != HJR_MODE_DEFAULT) { fprintf(stderr, "henjou_cli: the multi-GPU path renders Render_mode \"Default\" only\n"); return 1; }
    hjr_scene* scene = nullptr;
    HJRX(hjr_scene_load_gltf(opt.gltf_path, opt.gltf_name, &opt, &scene));
    hjr_scene_view view;
    HJR_INIT(view);
    HJRX(hjr_scene_get_view(scene, &view));
    hjr_ctx* ctx = nullptr;
    HJRX(hjr_create(rank, &ctx));
    HJRX(hjr_upload_scene(ctx, &view));
    {
        uint8_t* lut = nullptr; int lw = 0, lh = 0;
        if (hjr_load_png_rgba8(opt.LUT_path, &lut, &lw, &lh) == HJR_OK) { HJRX(hjr_set_lut(ctx, lut, lw, lh)); hjr_free(lut); }
        if (opt.use_IBL) {
            float* sky = nullptr; int sw = 0, sh = 0;
            if (hjr_load_hdr_rgba32f(opt.IBL_path, &sky, &sw, &sh) == HJR_OK) { HJRX(hjr_set_sky(ctx, sky, sw, sh)); hjr_free(sky); }
        }
    }
    // RCCL communicator: rank 0 creates the id and writes it to its pipe; the launcher passes it on to the pipes of the other ranks
    // (a world of one needs no hand-over).  Everything fallible that does not need the peers has happened above.
    HIPX(hipSetDevice(rank));
    ncclUniqueId id;
    if (rank == 0) {
        NCCLX(ncclGetUniqueId(&id));
        if (world > 1 && !io_all(id_fd, &id, sizeof(id), true)) { fprintf(stderr, "henjou_cli[0]: cannot hand the RCCL id to the launcher\n"); return 1; }
    } else if (!io_all(id_fd, &id, sizeof(id), false)) { fprintf(stderr, "henjou_cli[%d]: no RCCL id from rank 0\n", rank); return 1; }
    if (id_fd >= 0) close(id_fd);
    ncclComm_t comm;
    NCCLX(ncclCommInitRank(&comm, world, id, rank));
    hipStream_t st;
    HIPX(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));

    const uint32_t W = opt.image_width, H = opt.image_height;
    const size_t block = (size_t)hjr_owned_tiles(W, H, 0, (uint32_t)world) * 64; // float4 per rank (rank 0 owns the most tiles; others pad)
    float *d_packed = nullptr, *d_all = nullptr, *d_frame = nullptr;
    HIPX(hipMalloc(&d_packed, block * 16));
    HIPX(hipMemset(d_packed, 0, block * 16));
    std::vector<float> frame;
    if (rank == 0) {
        HIPX(hipMalloc(&d_all, block * 16 * world));
        HIPX(hipMalloc(&d_frame, (size_t)W * H * 16));
        frame.resize((size_t)W * H * 4);
    }
    std::vector<float> m((size_t)view.n_instances * 12), inv((size_t)view.n_instances * 12);
    for (uint32_t f = opt.start_frame; f < opt.end_frame; f++) {
        const float time = f / float(opt.fps); // renderer.h:1128
        HJRX(hjr_scene_eval_transforms(scene, time, m.data(), inv.data()));
        HJRX(hjr_set_transforms(ctx, m.data(), inv.data(), view.n_instances));
        hjr_params p;
        HJR_INIT(p);
        p.width = W; p.height = H; p.spp = opt.max_spp; p.frame = f; p.seed = opt.seed; p.integrator = (uint32_t)opt.integrator;
        HJRX(hjr_scene_eval_camera(scene, &opt, time, &p.camera));
        for (int k = 0; k < 3; k++) p.sky[k] = opt.scene_sky_default[k];
        p.ibl_intensity = opt.IBL_intensity;
        p.rank = (uint32_t)rank; p.world_size = (uint32_t)world; p.flags = HJR_FLAG_PACKED | (opt.fast_math ? HJR_FLAG_FAST_MATH : 0u);
        const auto t0 = std::chrono::steady_clock::now();
        HJRX(hjr_render_device(ctx, &p, d_packed, nullptr, nullptr, st));
        NCCLX(ncclGather(d_packed, d_all, block * 4, ncclFloat, 0, comm, st)); // the one data-path collective of a frame
        if (rank == 0) {
            for (int r = 0; r < world; r++) HJRX(hjr_unpack_tiles_device(ctx, d_all + (size_t)r * block * 4, W, H, (uint32_t)r, (uint32_t)world, d_frame, st));
            HIPX(hipMemcpyAsync(frame.data(), d_frame, frame.size() * 4, hipMemcpyDeviceToHost, st));
        }
        HIPX(hipStreamSynchronize(st));
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        hjr_stats s;
        HJR_INIT(s);
        if (hjr_get_stats(ctx, &s) == HJR_OK)
            fprintf(stderr, "[henjou %d/%d] frame %u: kernel %.3f ms, render + gather + assemble %.3f ms\n", rank, world, f, s.last_kernel_ms, ms);
        if (rank == 0) {
            std::vector<uint8_t> rgba8((size_t)W * H * 4);
            HJRX(hjr_float4_to_srgb8(frame.data(), rgba8.data(), W * H));
            std::string n = std::to_string(f); // renderer.h:1291-1302
            while (n.size() < 3) n = "0" + n;
            HJRX(hjr_write_png((std::string(opt.image_name) + "_" + n + ".png").c_str(), rgba8.data(), W, H, 1));
        }
    }
    ncclCommDestroy(comm);
    hjr_destroy(ctx);
    hjr_scene_free(scene);
    return 0;
}

int main(int argc, char** argv)
{
    const char* path = "render_option.json";
    int device = 0, devices = 0, rank = -1, world = 0, id_fd = -1;
    int positional = 0;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        if (a == "--devices" && i + 1 < argc) devices = atoi(argv[++i]);
        else if (a == "--rank" && i + 1 < argc) rank = atoi(argv[++i]);
        else if (a == "--world" && i + 1 < argc) world = atoi(argv[++i]);
        else if (a == "--id-fd" && i + 1 < argc) id_fd = atoi(argv[++i]);
        else if (positional == 0) { path = argv[i]; positional++; }
        else if (positional == 1) { device = atoi(argv[i]); positional++; }
    }
    // a rank process of a multi-GPU render (world 1: the same code path on one GPU, no id hand-over)
    if (rank >= 0) return (world >= 1 && rank < world && (world == 1 || id_fd >= 0)) ? run_rank(path, rank, world, id_fd) : 2;
    if (devices == 0) { // the JSON decides (host-only parse: no GPU is touched here)
        hjr_render_option opt;
        HJR_INIT(opt);
        if (hjr_load_render_option(path, &opt) != HJR_OK) { fprintf(stderr, "henjou_cli: error: %s\n", hjr_last_error()); return 1; }
        devices = (int)opt.devices;
    }
    if (devices <= 1) {
        int rc = hjr_render_file(path, device);
        if (rc != HJR_OK) { fprintf(stderr, "henjou_cli: error %d: %s\n", rc, hjr_last_error()); return 1; }
        return 0;
    }
    // launcher: one rank process per GPU, started before any GPU call in this process.  Pipes: up[0] <- rank 0 (the id), down[r] -> rank r.
    signal(SIGPIPE, SIG_IGN); // a rank that died before reading its id must not take the launcher down
    int up[2] = { -1, -1 };
    if (pipe(up) != 0) { perror("henjou_cli: pipe"); return 1; }
    std::vector<int> down((size_t)devices, -1);
    std::vector<pid_t> kids;
    auto kill_all = [&]() {
        for (pid_t k : kids) if (k > 0) kill(k, SIGTERM);
        for (int t = 0; t < 50; t++) { // 5 s of grace, then SIGKILL
            bool any = false;
            for (pid_t& k : kids) if (k > 0) { int stt; if (waitpid(k, &stt, WNOHANG) == k) k = -1; else any = true; }
            if (!any) return;
            std::this_thread::sleep_for(std::chrono::milliseconds(100));
        }
        for (pid_t& k : kids) if (k > 0) { kill(k, SIGKILL); int stt; waitpid(k, &stt, 0); k = -1; }
    };
    for (int r = 0; r < devices; r++) {
        int pd[2] = { -1, -1 };
        if (r > 0 && pipe(pd) != 0) { perror("henjou_cli: pipe"); kill_all(); return 1; }
        const pid_t pid = fork();
        if (pid < 0) { perror("henjou_cli: fork"); kill_all(); return 1; }
        if (pid == 0) {
            // the child keeps exactly one pipe end: rank 0 the write end of `up`, rank r the read end of its own `down` pipe
            close(up[0]);
            for (int q = 1; q < r; q++) if (down[(size_t)q] >= 0) close(down[(size_t)q]);
            int mine = up[1];
            if (r > 0) { close(up[1]); close(pd[1]); mine = pd[0]; }
            const std::string rs = std::to_string(r), ws = std::to_string(devices), fs = std::to_string(mine);
            execl("/proc/self/exe", argv[0], path, "--rank", rs.c_str(), "--world", ws.c_str(), "--id-fd", fs.c_str(), (char*)nullptr);
            perror("henjou_cli: exec");
            _exit(127);
        }
        kids.push_back(pid);
        if (r > 0) { close(pd[0]); down[(size_t)r] = pd[1]; }
    }
    close(up[1]);
    // wait for rank 0's id and for children at the same time: the first child that fails ends the job
    int rc = 0, alive = devices;
    bool id_sent = false;
    ncclUniqueId id;
    size_t got = 0;
    while (alive > 0 && rc == 0) {
        if (!id_sent) {
            struct pollfd pf = { up[0], POLLIN, 0 };
            if (poll(&pf, 1, 100) > 0) {
                const ssize_t k = read(up[0], (char*)&id + got, sizeof(id) - got);
                if (k > 0) got += (size_t)k;
                else if (k == 0 && got < sizeof(id)) { /* rank 0 closed its end without an id: its exit status follows */ std::this_thread::sleep_for(std::chrono::milliseconds(50)); }
                if (got == sizeof(id)) {
                    for (int r = 1; r < devices; r++) { (void)io_all(down[(size_t)r], &id, sizeof(id), true); close(down[(size_t)r]); down[(size_t)r] = -1; } // (a dead reader shows up as a failed child below)
                    id_sent = true;
                }
            }
        } else std::this_thread::sleep_for(std::chrono::milliseconds(50));
        for (;;) {
            int stt = 0;
            const pid_t k = waitpid(-1, &stt, WNOHANG);
            if (k <= 0) break;
            for (pid_t& q : kids) if (q == k) { q = -1; alive--; }
            if (!WIFEXITED(stt) || WEXITSTATUS(stt) != 0) {
                fprintf(stderr, "henjou_cli: a rank process %s; stopping the others\n", WIFEXITED(stt) ? "failed" : "was killed by a signal");
                rc = 1;
            }
        }
    }
    if (rc != 0) kill_all();
    close(up[0]);
    for (int fd : down) if (fd >= 0) close(fd);
    return rc;
}
