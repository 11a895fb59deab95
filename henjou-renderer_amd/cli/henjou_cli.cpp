// henjou_cli <render_option.json> [device] — stands in for the reference's missing main()
// (HenjouRenderer/henjouRenderer.cpp: Renderer r; r.initializeAndRender(path)).
#include <cstdio>
#include <cstdlib>

#include "../../include/henjou_hip.h"

int main(int argc, char** argv)
{
    const char* path = argc > 1 ? argv[1] : "render_option.json";
    int device = argc > 2 ? atoi(argv[2]) : 0;
    int rc = hjr_render_file(path, device);
    if (rc != HJR_OK) {
        fprintf(stderr, "henjou_cli: error %d: %s\n", rc, hjr_last_error());
        return 1;
    }
    return 0;
}
