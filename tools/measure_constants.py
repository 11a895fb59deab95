#!/usr/bin/env python3
"""Counter-measured per-sample constants of the roofline formula for the BASELINE.json configs, from the
kernel's counting variant (HJR_FLAG_STATS) on the GPU.  Prints a markdown table."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from scene_util import Cornell, hjr, load_lut  # noqa: E402

rows = [("C1 cornelbox 256x256x16", "render_option_c1.json", 256, 256, 16, False),
        ("C2 cornelbox 1920x1080 (16 of 256 spp)", "render_option_c2.json", 1920, 1080, 16, False),
        ("C2' diffuse/specular-only variant", "render_option_c2_nodiel.json", 1920, 1080, 16, False),
        ("C3 thin-film LUT (16 of 1024 spp)", "render_option_c3.json", 1920, 1080, 16, True),
        ("C4 negative-IOR glass, ior 1.5 (16 of 1024 spp)", "render_option_c4.json", 1920, 1080, 16, False)]
print("| config | R_c | R_s | box tests / closest ray | tri tests / closest ray | box tests / shadow ray | tri tests / shadow ray | H | S | B_sample [B] |")
print("|---|---|---|---|---|---|---|---|---|---|")
for name, cfg, w, h, spp, lut in rows:
    s = Cornell(cfg)
    d = s.device()
    if lut:
        d.set_lut(load_lut())
    d.render(s.hjr_params(w, h, spp, flags=hjr.FLAG_STATS), want_aovs=False)
    st = d.stats()
    d.close()
    n = st["samples"]
    rc, rs = st["closest_rays"] / n, st["shadow_rays"] / n
    b = (st["box_tests_closest"] * 32 + st["tri_tests_closest"] * 36 + st["box_tests_shadow"] * 32 + st["tri_tests_shadow"] * 36 +
         st["shaded_hits"] * 232 + st["light_samples"] * 192) / n + 52.0 / spp
    print("| %s | %.3f | %.3f | %.2f | %.2f | %.2f | %.2f | %.3f | %.3f | %.0f |" % (
        name, rc, rs, st["box_tests_closest"] / max(st["closest_rays"], 1), st["tri_tests_closest"] / max(st["closest_rays"], 1),
        st["box_tests_shadow"] / max(st["shadow_rays"], 1), st["tri_tests_shadow"] / max(st["shadow_rays"], 1),
        st["shaded_hits"] / n, st["light_samples"] / n, b))
