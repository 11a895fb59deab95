// Multi-process check (CPU only) of the pixel-tile shard that henjou_cli and bench.py run over RCCL: N rank processes each pack
// the tiles they own of the same synthetic frame (hjr_pack_tiles: exactly what HJR_FLAG_PACKED makes the kernel produce), hand the
// blocks to rank 0 through pipes (standing in for ncclGather), and rank 0 scatters them (hjr_unpack_tiles) and must get the frame
// back bit for bit, every pixel written exactly once.   usage: tile_shard_test <width> <height> <ranks>
#include <sys/wait.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/henjou_hip.h"

static float pixel(uint32_t x, uint32_t y, int ch) { return (float)((x * 131u + y * 7919u + (uint32_t)ch * 17u) % 100003u) * 0.25f + (float)ch; }

int main(int argc, char** argv)
{
    if (argc < 4) return 2;
    const uint32_t W = (uint32_t)atoi(argv[1]), H = (uint32_t)atoi(argv[2]), N = (uint32_t)atoi(argv[3]);
    std::vector<float> frame((size_t)W * H * 4);
    for (uint32_t y = 0; y < H; y++) for (uint32_t x = 0; x < W; x++) for (int c = 0; c < 4; c++) frame[((size_t)y * W + x) * 4 + c] = pixel(x, y, c);
    const size_t block = (size_t)hjr_owned_tiles(W, H, 0, N) * 64 * 4; // floats per rank: rank 0 owns the most tiles, the others pad
    uint64_t total_tiles = 0;
    for (uint32_t r = 0; r < N; r++) total_tiles += hjr_owned_tiles(W, H, r, N);
    if (total_tiles != (uint64_t)((W + 7) / 8) * ((H + 7) / 8)) { fprintf(stderr, "tiles are not partitioned\n"); return 1; }
    std::vector<int> rd(N);
    std::vector<pid_t> kids;
    for (uint32_t r = 0; r < N; r++) {
        int fd[2];
        if (pipe(fd) != 0) return 1;
        const pid_t pid = fork();
        if (pid == 0) { // rank r: pack the owned tiles and send the (padded) block
            close(fd[0]);
            std::vector<float> packed(block, -1.0f);
            if (hjr_pack_tiles(frame.data(), W, H, r, N, packed.data()) != HJR_OK) _exit(3);
            const char* p = (const char*)packed.data();
            size_t left = block * 4;
            while (left) { ssize_t k = write(fd[1], p, left); if (k <= 0) _exit(4); p += k; left -= (size_t)k; }
            _exit(0);
        }
        close(fd[1]);
        rd[r] = fd[0];
        kids.push_back(pid);
    }
    std::vector<float> out((size_t)W * H * 4, -7.0f), blk(block);
    for (uint32_t r = 0; r < N; r++) {
        char* p = (char*)blk.data();
        size_t left = block * 4;
        while (left) { ssize_t k = read(rd[r], p, left); if (k <= 0) { fprintf(stderr, "short read from rank %u\n", r); return 1; } p += k; left -= (size_t)k; }
        // before scattering, every pixel this rank owns must still be untouched (no tile belongs to two ranks)
        std::vector<float> probe = out;
        if (hjr_unpack_tiles(blk.data(), W, H, r, N, out.data()) != HJR_OK) return 1;
        for (size_t i = 0; i < out.size(); i++) if (out[i] != probe[i] && probe[i] != -7.0f) { fprintf(stderr, "pixel written twice\n"); return 1; }
    }
    int rc = 0;
    for (pid_t k : kids) { int st = 0; waitpid(k, &st, 0); if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) rc = 1; }
    if (memcmp(out.data(), frame.data(), frame.size() * 4) != 0) { fprintf(stderr, "assembled frame differs\n"); rc = 1; }
    if (rc == 0) printf("tile_shard_test ok: %ux%u over %u ranks, %zu floats per block\n", W, H, N, block);
    return rc;
}
