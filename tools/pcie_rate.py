"""PCIe-inclusive cost of the host-buffer entry point hjr_render (what the reference's boundary hands over) on C2: wall time of the
call vs the kernel time inside it.  Run on the GPU box from the repo root."""
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import __graft_entry__ as entry
hjr = entry.load_package()
from scene_util import Cornell
s = Cornell("render_option_c2.json")
d = s.device()
p = s.hjr_params(1920, 1080, 256)
for mode in ("color only", "color+albedo+normal"):
    for i in range(3):
        t0 = time.perf_counter()
        if mode == "color only":
            out = d.render(p, want_aovs=False)
        else:
            out = d.render(p)
        dt = (time.perf_counter() - t0) * 1e3
        k = d.stats()["last_kernel_ms"]
        print(mode, "hjr_render wall %.1f ms, kernel %.1f ms, overhead %.1f ms" % (dt, k, dt - k))
d.close()
